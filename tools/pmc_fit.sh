#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fit
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
CMD="python $R/bench_fit.py --iters 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ('p1','p2'):
    tot=collections.defaultdict(collections.Counter); n=collections.defaultdict(collections.Counter)
    for path in glob.glob('$O/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(path)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
            if k.startswith('k_fit'):
                tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    for k in tot:
        print(d, k, {c: tot[k][c]/n[k][c] for c in tot[k]})
PY
