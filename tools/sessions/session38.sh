#!/bin/bash
# GPU session 38: per-bounce launch times on ONE stream (no kernel beside another): how efficient are the last bounces by themselves?
set -o pipefail
out=$PWD/gpurun_out/s38; mkdir -p $out; root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --config 1 --no-cpu-baseline --overlap 0 > $out/bench_one_stream.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 tools/per_bounce.py $(find $out/kt -name "*kernel_trace.csv" | head -1) > $out/per_bounce_one_stream.json && cat $out/per_bounce_one_stream.json
rm -rf $out/kt
