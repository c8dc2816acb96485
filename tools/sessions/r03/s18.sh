#!/bin/bash
# round 3, GPU session 18: the next batch's raygen on its own stream (ptmi_options.pipeline = 2): parity, then interleaved A/B where it
# can act — consecutive asynchronous dispatches (config 1 and 3 with --steps 8, the driver runs 20) and multi-batch dispatches (configs 2, 4)
set -o pipefail
out=gpurun_out/r03_s18; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_state.py -m gpu -x -q > $out/pytest_state.log 2>&1; rc=$?; tail -5 $out/pytest_state.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('pipeline_used'))"; }
for i in 1 2 3; do
  run c1_off_$i --config 1 --steps 8 --pipeline 1 && run c1_on_$i --config 1 --steps 8 --pipeline 2 || exit 1
done
for i in 1 2; do
  run c3_off_$i --config 3 --steps 6 --pipeline 1 && run c3_on_$i --config 3 --steps 6 --pipeline 2 || exit 1
  run c4_off_$i --config 4 --pipeline 1 && run c4_on_$i --config 4 --pipeline 2 || exit 1
  run c2_off_$i --config 2 --pipeline 1 && run c2_on_$i --config 2 --pipeline 2 || exit 1
done
run c1_off_1step --config 1 --pipeline 1 && run c1_on_1step --config 1 --pipeline 2
