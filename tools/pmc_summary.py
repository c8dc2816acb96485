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
def to_json(paths, commit=None):
    """{counter: {kernel: {launches, avg_KB_per_launch}}} — the form committed as profiles/rNN_bench_n1_pmc.json.
    The two traversal kernels share a template name; they are told apart by their IO type."""
    import json
    out = collections.defaultdict(dict)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r['Kernel_Name']
            m = re.search(r'(k_\w+)', name)
            if not m:
                continue
            k = m.group(1)
            if k.startswith(('k_trace', 'k_own')):
                k += '/shadow' if 'ShadowIO' in name else '/extend'
            a = acc[(r['Counter_Name'], k)]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    for (c, k), (n, s) in sorted(acc.items()):
        out[c][k] = {"launches": n, ("avg_KB_per_launch" if c.endswith("_SIZE") else "avg_per_launch"): round(s / n, 2)}
    if commit:
        out["_code_commit"] = commit        # the commit of the code these counters measured (the GPU box has no .git)
    return json.dumps(out, indent=1)


if __name__ == '__main__' and sys.argv[1:2] == ['--json']:
    rest, commit = sys.argv[2:], None
    if rest[:1] == ['--commit']:
        commit, rest = rest[1], rest[2:]
    print(to_json(sum([glob.glob(p) for p in rest], []), commit))
elif __name__ == '__main__':
    agg, meta = summarise(sum([glob.glob(p) for p in sys.argv[1:]], []))
    for k in sorted(agg):
        print(k, 'vgpr/agpr/sgpr/lds/wg/scratch =', meta[k])
        for c, (n, s) in sorted(agg[k].items()):
            print(f'   {c:28s} launches={n:4d} sum={s:.4g} per_launch={s/n:.4g}')

