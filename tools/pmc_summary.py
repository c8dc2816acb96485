#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, sum and mean per launch."""
import collections, csv, glob, re, sys
def summarise(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    meta = {}
    for path in paths:
        for r in csv.DictReader(open(path)):
            m = re.search(r'(k_\w+)', r['Kernel_Name'])
            if not m: continue
            k = m.group(1)
            a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
            meta[k] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size'], r['Scratch_Size'])
    return agg, meta
if __name__ == '__main__':
    agg, meta = summarise(sum([glob.glob(p) for p in sys.argv[1:]], []))
    for k in sorted(agg):
        print(k, 'vgpr/agpr/sgpr/lds/wg/scratch =', meta[k])
        for c, (n, s) in sorted(agg[k].items()):
            print(f'   {c:28s} launches={n:4d} sum={s:.4g} per_launch={s/n:.4g}')
