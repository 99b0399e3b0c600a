#!/usr/bin/env python
"""Per-kernel averages of the SQ counter passes written by tools/pmc_sq.sh."""
import collections
import csv
import glob
import json
import sys

KERNELS = ('k_d4c_body', 'k_d4c_bands', 'k_cheaptrick', 'k_d4c_lovetrain', 'k_syn_pulse', 'k_sp2mc', 'k_mc2sp_mfma')


def main(root):
    tot = collections.defaultdict(collections.Counter)
    n = collections.defaultdict(collections.Counter)
    for path in glob.glob(root + '/pass*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            base = r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '').strip()
            if base in KERNELS:
                tot[base][r['Counter_Name']] += float(r['Counter_Value'])
                n[base][r['Counter_Name']] += 1
    out = {}
    for k in KERNELS:
        if k not in tot:
            continue
        c = {name: tot[k][name] / n[k][name] for name in tot[k]}
        d = dict(c)
        wc = c.get('SQ_WAVE_CYCLES')
        if wc:
            for name in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
                         'SQ_ACTIVE_INST_LDS'):
                if name in c:
                    d[name + '/WAVE_CYCLES'] = c[name] / wc
        if c.get('SQ_LDS_IDX_ACTIVE'):
            d['LDS_BANK_CONFLICT/LDS_IDX_ACTIVE'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']
        if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_ACTIVE_INST_VALU'):
            # SQ_ACTIVE_INST_* count quad-cycles summed over all waves; GRBM_GUI_ACTIVE is summed over the
            # 8 XCDs.  Share of the launch during which a SIMD (4 x 256 of them) issues a vector instruction:
            d['VALU_busy_per_SIMD'] = c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (c['GRBM_GUI_ACTIVE'] / 8)
        if c.get('SQ_WAVES') and c.get('SQ_INSTS_VALU'):
            d['VALU_insts_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
        out[k] = d
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
