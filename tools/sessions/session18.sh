#!/bin/bash
# GPU session 18: per-bounce kernel times on one stream, Cornell against the enclosed Cornell (dense bounce-1 queue)
set -o pipefail
out=gpurun_out/s18; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for sc in cornell cornell_enclosed; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$sc -- python3 bench.py --config 1 --scene $sc --overlap 0 --no-cpu-baseline > $out/bench_$sc.json 2> $out/kt_$sc.err || exit 1
python3 tools/per_bounce.py $(find $out/kt_$sc -name "*kernel_trace.csv" | head -1) > $out/per_bounce_$sc.json && cat $out/per_bounce_$sc.json
python3 -c "
import json; d=json.load(open('$out/bench_$sc.json')); print('$sc', d['value'], d['kernel_ms_rank0'], d['segments_by_bounce_rank0'])"
rm -rf $out/kt_$sc
done
