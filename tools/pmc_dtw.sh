#!/bin/bash
# SQ counter passes for the FastDTW recurrence kernel (run on the GPU box through gpurun): instruction counts by kind,
# issue and wait cycles of the finest level's launch of one 2201 x 2401 call (tools/dtw_run.py).
R=$GRAFT_REPO_ROOT; O=${KWY_MEASURE_OUT:-$R/gpurun_out/pmc_dtw}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/pass1 -- python $R/tools/dtw_run.py > $O/pass1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $O/pass2 -- python $R/tools/dtw_run.py > $O/pass2.log 2>&1
python - <<PY > $O/summary.json
import csv, glob, collections, json
out = {'command': 'rocprofv3 --pmc ... --kernel-trace -- python tools/dtw_run.py (two passes)', 'shape': '2201 x 2401 x 26, radius 32',
       'note': 'the finest level\'s launch (the largest) of every kernel; counters summed over the wavefronts of the launch'}
for kname in ('k_dtw_values', 'k_dtw_codes', 'k_dtw_trace', 'k_dtw_dist'):
    acc = {}
    for path in glob.glob('$O/pass*/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if kname in r['Kernel_Name']]
        by = collections.defaultdict(dict)
        for r in rows: by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        if not by: continue
        key = max(by, key=lambda d: max(by[d].values()))
        acc.update(by[key])
    out[kname] = acc
print(json.dumps(out, indent=1))
PY
echo done > $O/DONE
