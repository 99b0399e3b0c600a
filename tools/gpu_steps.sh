#!/bin/bash
# Run a list of GPU steps one after the other under a per-step timeout; a failing step (tests red) does not stop
# the list, a step that TIMES OUT or is killed does (no further GPU work after a hang).
# usage: tools/gpu_steps.sh "<secs> <logname> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "=== $name: $cmd" | tee gpurun_out/$name.log
  timeout -k 10 $secs bash -c "$cmd" >> gpurun_out/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/$name.log
  tail -n 4 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
done
exit 0
