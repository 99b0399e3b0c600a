#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -- python $R/bench.py --batch 1 --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_b1.log 2>&1
