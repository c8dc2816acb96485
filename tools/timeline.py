#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [n_last]: start / end (us, relative to the first listed launch) of the last n_last kernel
launches of a rocprofv3 --kernel-trace CSV of bench.py, with the stream (queue) each ran on — to see what runs beside what."""
import csv, re, sys


def main(path, n_last=70):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n_last:]
    t0 = int(rows[0]["Start_Timestamp"])
    seen = {}
    for r in rows:
        name = r["Kernel_Name"]
        m = re.search(r"(k_\w+)", name)
        k = m.group(1) if m else name[:24]
        if k.startswith(("k_trace", "k_own")):
            k = ("shadow" if "ShadowIO" in name else "extend") + ("_lds" if "lds" in k else "_glb")
        b = seen.get(k, 0); seen[k] = b + 1
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{k:16s} #{b:<3d} q{r.get('Queue_Id', '?'):>3s} {s:10.1f} {e:10.1f} {e - s:9.1f}  wg={r.get('Workgroup_Size', '?')} grid={r.get('Grid_Size', '?')} lds={r.get('LDS_Block_Size', '?')}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 70)
