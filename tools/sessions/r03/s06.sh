#!/bin/bash
# round 3, GPU session 6: the ray state follows the queue (ptmi_options.state = 2) — parity first, then interleaved same-box A/B
# against the state in place on configs 1 and 3 (per-bounce times by --timing 3 are in the JSON lines)
set -o pipefail
out=gpurun_out/r03_s06; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_state.py -m gpu -x -q > $out/pytest_state.log 2>&1; rc=$?; tail -5 $out/pytest_state.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('state_used'))"; }
for i in 1 2 3; do
  run c1_inplace_$i --config 1 --state 1 && run c1_follow_$i --config 1 --state 2 || exit 1
done
for i in 1 2; do
  run c3_inplace_$i --config 3 --state 1 && run c3_follow_$i --config 3 --state 2 || exit 1
done
run c1_inplace_one --config 1 --state 1 --overlap 0 && run c1_follow_one --config 1 --state 2 --overlap 0 || exit 1
run c2_inplace --config 2 --state 1 && run c2_follow --config 2 --state 2 || exit 1
