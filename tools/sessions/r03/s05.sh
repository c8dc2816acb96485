#!/bin/bash
# round 3, GPU session 5 (re-entry): the whole GPU suite on HEAD, then the default bench lines of configs 1 and 3
set -o pipefail
out=gpurun_out/r03_s05; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -8 $out/pytest_gpu.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python bench.py > $out/bench_cfg1.json 2> $out/bench_cfg1.err || { tail -5 $out/bench_cfg1.err; exit 1; }
timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err || { tail -5 $out/bench_cfg3.err; exit 1; }
python - <<'PY'
import json
for c in (1, 3):
    d = json.load(open(f'gpurun_out/r03_s05/bench_cfg{c}.json'))
    print(c, d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'])
PY
