#!/usr/bin/env python
"""How busy is the chip during the timed steps of `bench.py` (lockstep driver)?

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --no-variants --no-cpu-baseline
    python tools/step_trace.py DIR/*/*_kernel_trace.csv [steps] [warmup]

A step of the lockstep driver starts with ONE k_np_words launch (the generator's head, first kernel of the first wave's
side stream), so the steps are delimited by those launches.  Over the timed steps (the `steps` replays behind
1 + warmup passes) this prints: the step time from the trace, the share of it during which at least one WHOLE-CHIP kernel
(grid >= 256 workgroups) was executing, the share with only narrow kernels (serial recurrences, scans) executing, the idle
share, and per kernel its summed duration per step."""
import csv
import json
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '').strip()
            wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
            grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, grid // max(1, wg)))
    rows.sort()
    heads = [a for a, b, n, g in rows if n == 'k_np_words']
    # passes: the constructor draws nothing; capture() = one plain pass; warmup replays; `steps` timed replays
    first = 1 + warmup
    if len(heads) < first + steps + 1:
        raise SystemExit(f'{len(heads)} steps in the trace, need {first + steps + 1}')
    t0, t1 = heads[first], heads[first + steps]
    sel = [(a, b, n, g) for a, b, n, g in rows if a >= t0 and a < t1]
    wide = [(a, min(b, t1)) for a, b, n, g in sel if g >= 256]
    anyk = [(a, min(b, t1)) for a, b, n, g in sel]
    span = t1 - t0
    per = defaultdict(lambda: [0, 0, 0])
    for a, b, n, g in sel:
        per[n][0] += b - a
        per[n][1] += 1
        per[n][2] = max(per[n][2], g)
    out = {
        'trace': path, 'steps': steps, 'ms_per_step': span / steps / 1e6,
        'share_with_a_whole_chip_kernel_executing': union(wide) / span,
        'share_with_only_narrow_kernels_executing': (union(anyk) - union(wide)) / span,
        'share_idle': 1.0 - union(anyk) / span,
        'summed_kernel_ms_per_step': sum(v[0] for v in per.values()) / steps / 1e6,
        'kernels': {n: {'ms_per_step': v[0] / steps / 1e6, 'launches_per_step': v[1] / steps, 'max_workgroups': v[2]}
                    for n, v in sorted(per.items(), key=lambda kv: -kv[1][0])},
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
