#!/bin/bash
# tools/gpurun_wait.sh LOG TIMEOUT 'command'
# gpurun with a wait for a free slot: exit code 3 ("no box or slot free right now", nothing charged, the command did
# not start) is retried after two minutes, up to 20 times.  Any other outcome -- including a failed or killed command
# -- is returned as it is: a GPU command is never run twice.
log=$1; shift
to=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
