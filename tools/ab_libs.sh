#!/bin/bash
# A/B of two builds of libkwy.so on ONE GPU box (boxes differ by ~2 % in clocks: numbers of different gpurun calls
# do not compare to better than that).  Build both variants here, keep them as scratch/libkwy_old.so and
# scratch/libkwy_new.so (scratch/ is git-ignored but travels with gpurun), then
#   gpurun -- bash tools/ab_libs.sh            alternating runs old, new, old, new of the default bench
# and read kernel_ms_per_launch_alone / value from gpurun_out/ab/ab_{old,new}_{1,2}.json.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cp $R/kwiiyatta_amd/libkwy.so $R/scratch/libkwy_tree.so
for i in 1 2; do
  for v in old new; do
    cp $R/scratch/libkwy_$v.so $R/kwiiyatta_amd/libkwy.so
    timeout -k 10 300 python $R/bench.py --no-variants --config4 off --no-cpu-baseline > $O/ab_${v}_$i.json 2> $O/ab_${v}_$i.err || exit 1
  done
done
cp $R/scratch/libkwy_tree.so $R/kwiiyatta_amd/libkwy.so
