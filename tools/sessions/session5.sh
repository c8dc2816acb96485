#!/bin/bash
# GPU session 5: LDS top-of-tree cache of the quantised variant, perf mode, sharpened grazing test
set -o pipefail
out=gpurun_out/s5; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_golden.py tests/test_gpu_perf_mode.py -m gpu -x -q -s > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; grep -h "grazing rays\|perf mode" $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2; do
run cfg3_qc256_$i --config 3 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_qc128.so run cfg3_qc128_$i --config 3 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_qc1.so run cfg3_qc1_$i --config 3 &&
run cfg3_exact_$i --config 3 --traversal global_exact || exit 1
done
run cfg3_sort --config 3 --sort 1 &&
run cfg1_parity --config 1 && run cfg1_perf --config 1 --perf-mode 1 && run cfg1_parity2 --config 1 && run cfg1_perf2 --config 1 --perf-mode 1 &&
run cfg2_perf --config 2 --steps 4 --perf-mode 1 && run cfg2_parity --config 2 --steps 4 &&
bash tools/pmc_cfg.sh 3 cfg3 > $out/pmc_cfg3.log 2>&1; cat $out/pmc_cfg3.log
