#!/bin/bash
# GPU session 8: overlapped shadow stream: parity (whole suite), then A/B --overlap 0 / 1 on every config
set -o pipefail
out=gpurun_out/s8; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['kernel_ms_sum_over_gpu_ms'])"; }
for i in 1 2 3; do run cfg1_seq_$i --config 1 --overlap 0 && run cfg1_ovl_$i --config 1 --overlap 1 || exit 1; done
run cfg2_seq --config 2 --steps 4 --overlap 0 && run cfg2_ovl --config 2 --steps 4 --overlap 1 &&
run cfg3_seq --config 3 --overlap 0 && run cfg3_ovl --config 3 --overlap 1 && run cfg3_seq2 --config 3 --overlap 0 && run cfg3_ovl2 --config 3 --overlap 1 &&
run cfg4_seq --config 4 --steps 2 --overlap 0 && run cfg4_ovl --config 4 --steps 2 --overlap 1
