#!/bin/bash
# tools/clock_under_load.sh <config> <out.json>: effective engine clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
# (MI355X_MICROARCH.md, DVFS), from ONE rocprofv3 run with --pmc GRBM_GUI_ACTIVE --kernel-trace (kernels serialised by the profiler).
cfg=${1:-1}; outj=${2:-gpurun_out/clock_cfg$cfg.json}
root=$PWD; d=$root/gpurun_out/clk_$cfg; mkdir -p $d; cd /tmp; export TMPDIR=/tmp; cd $root
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 bench.py --config $cfg --no-cpu-baseline > $d/bench.json 2> $d/err.log || { tail -3 $d/err.log; exit 1; }
python3 - "$d" "$outj" <<'PY'
import csv, glob, json, re, sys, collections
d, outj = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    ns, name = dur[r["Dispatch_Id"]]
    m = re.search(r"(k_\w+)", name)
    if not m or ns < 300000:          # the quotient reads high on dispatches shorter than ~0.3 ms
        continue
    k = m.group(1) + ("/shadow" if "ShadowIO" in name else "/extend" if "ExtendIO" in name else "")
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
out = {k: {"launches_over_0.3ms": n, "ghz": round(c / 8.0 / ns, 3)} for k, (c, ns, n) in sorted(acc.items())}
json.dump(out, open(outj, "w"), indent=1); print(json.dumps(out))
PY
rm -rf $d
