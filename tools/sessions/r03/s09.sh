#!/bin/bash
# round 3, GPU session 9: the last bounces on one stream (ptmi_options.tails = 2): parity, then interleaved A/B on configs 1, 3, 2
set -o pipefail
out=gpurun_out/r03_s09; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_state.py -m gpu -x -q > $out/pytest_state.log 2>&1; rc=$?; tail -5 $out/pytest_state.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('tails_used'))"; }
for i in 1 2 3; do
  run c1_tails1_$i --config 1 --tails 1 && run c1_tails2_$i --config 1 --tails 2 || exit 1
done
for i in 1 2; do
  run c3_tails1_$i --config 3 --tails 1 && run c3_tails2_$i --config 3 --tails 2 || exit 1
done
run c2_tails1 --config 2 --tails 1 && run c2_tails2 --config 2 --tails 2 || exit 1
