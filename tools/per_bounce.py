#!/usr/bin/env python3
"""tools/per_bounce.py <kernel_trace.csv> [bounces]: per-bounce mean duration (us) of extend / shade / shadow from a
rocprofv3 --kernel-trace CSV of bench.py (launches of one kernel cycle through the bounces in order)."""
import collections, csv, json, re, sys


def main(path, bounces=8):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.Counter()
    for r in rows:
        name = r["Kernel_Name"]
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
        per[k][b].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {k: [round(sum(v[b]) / len(v[b]), 1) if v[b] else None for b in range(bounces)] for k, v in per.items()}
    print(json.dumps({"unit": "us per launch, mean over steps", "per_bounce": out}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
