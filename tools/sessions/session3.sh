#!/bin/bash
# GPU session 3: fused node test of the quantised variant (parity, then speed at 5 / 6 waves per SIMD), ray sort A/B
set -o pipefail
out=gpurun_out/s3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['upload_ms_rank0'])"; }
for i in 1 2; do
run cfg3_q5_$i --config 3 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_w6.so run cfg3_q6_$i --config 3 &&
run cfg3_exact_$i --config 3 --traversal global_exact || exit 1
done
for i in 1 2; do
run cfg1_nosort_$i --config 1 &&
run cfg1_sort_$i --config 1 --sort 1 || exit 1
done
run cfg2_nosort --config 2 --steps 4 && run cfg2_sort --config 2 --steps 4 --sort 1 &&
run cfg3_sort --config 3 --sort 1
