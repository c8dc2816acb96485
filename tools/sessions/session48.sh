#!/bin/bash
# GPU session 48: overlap 4 (the shadow kernel of bounce b held back until extend of bounce b + 1 has finished) — parity test of
# the stream modes, then A/B against overlap 1
set -o pipefail
out=gpurun_out/s48; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_math.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run cfg1_o1_$i --config 1 --overlap 1 && run cfg1_o4_$i --config 1 --overlap 4 || exit 1; done
for i in 1 2; do run cfg3_o1_$i --config 3 --overlap 1 && run cfg3_o4_$i --config 3 --overlap 4 || exit 1; done
run cfg2_o1 --config 2 --overlap 1 && run cfg2_o4 --config 2 --overlap 4
