#!/bin/bash
# the timed steps of the default bench under rocprofv3 (run on the GPU box through gpurun):
#   kernel trace of the lockstep step -> tools/step_trace.py; --stats of the one-stream driver (the launch durations
#   bench.py's roofline is priced on)
R=$GRAFT_REPO_ROOT
O=${KWY_MEASURE_OUT:-$R/gpurun_out/prof_step}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-variants --no-cpu-baseline > $O/trace.log 2>&1
python $R/tools/step_trace.py "$(ls -t $O/trace/*/*kernel_trace.csv | head -1)" 20 25 > $O/step_busy.json
rm -f $O/trace/*/*kernel_trace.csv          # (hundreds of MB; the summary is what is kept)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/bench.py --driver serial --no-variants --no-cpu-baseline > $O/serial.log 2>&1
cp "$(ls -t $O/serial/*/*kernel_stats.csv | head -1)" $O/serial_kernel_stats.csv
rm -f $O/serial/*/*kernel_trace.csv
echo done > $O/DONE
