#!/usr/bin/env python
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only)
of  `bench.py --workload utterance --batch 1 --steps 2 --warmup 1 --no-cpu-baseline`
into per-launch HBM traffic of the frame kernels:  profiles/<round>_pmc_traffic.json.

    python profiles/make_pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv [round, default r2]

FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section).  That guide calibrates
FETCH_SIZE only for 16 B/lane streaming reads (it then shows 1/2 of the bytes); these kernels read
8 B/lane plus small tables, so the raw sum is what bench.py reports as `traffic` and the
fetch-doubled sum is kept beside it.
"""
import collections
import csv
import json
import os
import sys

KERNELS = ('k_d4c_body', 'k_d4c_bands', 'k_cheaptrick', 'k_d4c_lovetrain', 'k_syn_pulse', 'k_syn_ola')
FRAMES = 2001


def per_launch(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0]
        if 'k_syn_pulse' in name and name.rstrip().endswith(', true>'):      # k_syn_pulse<N, true>: the (normally empty) overflow pass
            continue
        base = name.split('<')[0].replace('void ', '').strip()
        if base in KERNELS:     # exact kernel name (k_d4c_body, not k_d4c_body_counts)
            tot[base] += float(r['Counter_Value'])
            n[base] += 1
    return {k: tot[k] / n[k] for k in KERNELS if n[k]}


def main():
    fetch, write = per_launch(sys.argv[1], 'FETCH_SIZE'), per_launch(sys.argv[2], 'WRITE_SIZE')
    out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (two separate passes) -- python bench.py '
                      '--workload utterance --batch 1 --steps 2 --warmup 1 --no-cpu-baseline',
           'frames_per_launch': FRAMES,
           'note': 'KB per launch; raw = (FETCH+WRITE)*1024, fetch_doubled = (2*FETCH+WRITE)*1024 '
                   '(gfx950 FETCH_SIZE halves wide streaming reads; uncalibrated for these 8 B/lane kernels)',
           'kernels': {}}
    for k in KERNELS:
        if k in fetch and k in write:
            out['kernels'][k] = {'FETCH_SIZE_KB_per_launch': fetch[k], 'WRITE_SIZE_KB_per_launch': write[k],
                                 'hbm_bytes_per_launch_raw': (fetch[k] + write[k]) * 1024,
                                 'hbm_bytes_per_launch_fetch_doubled': (2 * fetch[k] + write[k]) * 1024}
    here = os.path.dirname(os.path.abspath(__file__))
    rnd = sys.argv[3] if len(sys.argv) > 3 else 'r4'
    if len(sys.argv) > 4:          # frames per analysis launch of the profiled command (r4: 16 utterances = 33 616)
        out['frames_per_launch'] = int(sys.argv[4])
        out['command'] = sys.argv[5] if len(sys.argv) > 5 else out['command']
    json.dump(out, open(os.path.join(here, f'{rnd}_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
