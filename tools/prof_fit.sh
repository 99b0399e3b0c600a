#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_fit
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench_fit.py > $O/log.txt 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
