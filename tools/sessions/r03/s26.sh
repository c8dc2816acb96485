#!/bin/bash
# round 3, GPU session 26: which hardware queues do a SECOND context's main and shadow streams get (tools/two_contexts.py b: the first
# context destroyed before the second is made — measured 9 % slower)? default build and the shadow stream at high priority
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s26; mkdir -p $out
for v in def sidehigh; do
  lib=$root/wgpu-path-tracing_amd/lib/libptmi.so; [ $v != def ] && lib=$root/wgpu-path-tracing_amd/lib/ab/libptmi_$v.so
  export PTMI_LIB=$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_$v -- python3 tools/two_contexts.py b > $out/run_$v.txt 2> $out/run_$v.err || { tail -3 $out/run_$v.err; exit 1; }
  tail -1 $out/run_$v.txt
  f=$(find $out/kt_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
for part, rs in (("first context (first half of the launches)", rows[: n // 2]), ("second context (second half)", rows[n // 2:])):
    c = collections.Counter()
    for r in rs:
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:20]
        if k.startswith("k_trace"): k = "shadow" if "ShadowIO" in r["Kernel_Name"] else "extend"
        c[(k, r["Queue_Id"])] += 1
    print(part, dict(c))
PY
  rm -rf $out/kt_$v
done
