#!/usr/bin/env python3
"""tools/pmc_per_bounce.py [--bounces 8] <counter_collection.csv ...>: per-bounce means of every counter for extend / shade /
shadow from rocprofv3 --pmc CSVs of bench.py (the launches of one kernel cycle through the bounces in dispatch order)."""
import collections, csv, glob, json, re, sys


def main(paths, bounces):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [[0, 0.0] for _ in range(bounces)]))
    for path in paths:
        rows = collections.defaultdict(dict)               # dispatch id -> {counter: value}, kernel name
        names = {}
        for r in csv.DictReader(open(path)):
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[d] = r["Kernel_Name"]
        seen = collections.Counter()
        for d in sorted(rows):
            name = names[d]
            m = re.search(r"(k_\w+)", name)
            if not m:
                continue
            k = m.group(1)
            if k.startswith(("k_trace", "k_own")):
                k = "shadow" if "ShadowIO" in name else "extend"
            elif k == "k_shade":
                k = "shade"
            else:
                continue
            b = seen[k] % bounces
            seen[k] += 1
            for c, v in rows[d].items():
                a = out[c][k][b]; a[0] += 1; a[1] += v
    res = {c: {k: [round(s / n, 1) if n else None for n, s in v] for k, v in ks.items()} for c, ks in sorted(out.items())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    args = sys.argv[1:]
    nb = 8
    if args[:1] == ["--bounces"]:
        nb, args = int(args[1]), args[2:]
    main(sum([glob.glob(p) for p in args], []), nb)
