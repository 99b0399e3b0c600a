#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dtw_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/tools/dtw_run.py > $O/log.txt 2>&1
python - <<PY
import csv, glob
rows = []
for path in glob.glob('$O/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the second call only
names = [r['Kernel_Name'].split('(')[0].replace('void ', '') for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_dtw_halve_all')]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r, n in list(zip(rows, names))[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-28s start %8.1f us  dur %7.1f us  gap %5.1f' % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
PY
