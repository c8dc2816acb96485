#!/bin/bash
# GPU session 14: two half-batches in flight (overlap 3): whole suite, then A/B against overlap 1 on every config
set -o pipefail
out=gpurun_out/s14; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run cfg1_o1_$i --config 1 --overlap 1 && run cfg1_o3_$i --config 1 --overlap 3 || exit 1; done
run cfg1_o3_8steps --config 1 --overlap 3 --steps 8 && run cfg1_o1_8steps --config 1 --overlap 1 --steps 8 &&
run cfg3_o1 --config 3 --overlap 1 && run cfg3_o3 --config 3 --overlap 3 && run cfg3_o1b --config 3 --overlap 1 && run cfg3_o3b --config 3 --overlap 3 &&
run cfg2_o1 --config 2 --steps 4 --overlap 1 && run cfg2_o3 --config 2 --steps 4 --overlap 3 &&
run cfg4_o1 --config 4 --steps 2 --overlap 1 && run cfg4_o3 --config 4 --steps 2 --overlap 3
