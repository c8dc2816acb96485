#!/bin/bash
# GPU session 2: parity tests of the quantised global variant, then A/B on the scenes that use it
set -o pipefail
out=gpurun_out/s2; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; [ $rc = 0 ] || exit $rc
python tools/time_upload.py grid_1m > $out/upload.log 2>&1 && cat $out/upload.log &&
for t in global_exact auto; do for c in 3 2; do extra=""; [ $c = 2 ] && extra="--steps 4 --traversal $t" || extra="--traversal $t"; [ $c = 2 ] && [ $t = auto ] && extra="--steps 4 --traversal global"
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline $extra > $out/bench_cfg${c}_$t.json 2> $out/bench_cfg${c}_$t.err || exit 1; python -c "
import json; d=json.load(open('$out/bench_cfg${c}_$t.json')); print('cfg$c $t', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['upload_ms_rank0'], d['config']['traversal'])"; done; done
