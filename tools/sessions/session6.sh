#!/bin/bash
# GPU session 6: per-bounce kernel times (kernel trace) for configs 1 and 3, lane utilisation for config 1
set -o pipefail
out=gpurun_out/s6; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for c in 1 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$c -- python3 bench.py --config $c --no-cpu-baseline > $out/bench_cfg$c.json 2> $out/kt$c.err || exit 1
python3 tools/per_bounce.py $(find $out/kt$c -name "*kernel_trace.csv" | head -1) > $out/per_bounce_cfg$c.json && cat $out/per_bounce_cfg$c.json
python3 -c "
import json; d=json.load(open('$out/bench_cfg$c.json')); print('cfg$c', d['value'], d['kernel_ms_rank0']); print(' segments by bounce (rank 0 stats not in line)')"
rm -rf $out/kt$c
done
bash tools/pmc_cfg.sh 1 cfg1 > $out/pmc_cfg1.log 2>&1; cat $out/pmc_cfg1.log
