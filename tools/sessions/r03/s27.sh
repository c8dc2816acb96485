#!/bin/bash
# round 3, GPU session 27: second-context slowdown against the shadow stream's priority: modes a (two contexts alive) and b (first destroyed)
# of tools/two_contexts.py for the default (normal), low and high priority builds, with the hardware queue of main / shadow kernels
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s27; mkdir -p $out
for v in def sidelow sidehigh; do
  lib=$root/wgpu-path-tracing_amd/lib/libptmi.so; [ $v != def ] && lib=$root/wgpu-path-tracing_amd/lib/ab/libptmi_$v.so
  export PTMI_LIB=$lib
  for mode in a b; do
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 tools/two_contexts.py $mode > $out/run.txt 2> $out/run.err || { tail -3 $out/run.err; exit 1; }
    echo "$v $(tail -1 $out/run.txt)"
    f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
    python3 - "$f" <<'PY'
import csv, sys, collections, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
c = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    if "k_shade" in name: c[("main", r["Queue_Id"])] += 1
    elif "ShadowIO" in name: c[("shadow", r["Queue_Id"])] += 1
print("   queues:", dict(c))
PY
    rm -rf $out/kt
  done
done
