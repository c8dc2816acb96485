#!/bin/bash
# GPU session 50: shade fetching its next queue entry and hit record one iteration ahead — parity, then A/B
set -o pipefail
out=gpurun_out/s50; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_nopf.so run cfg1_plain_$i --config 1 && run cfg1_prefetch_$i --config 1 || exit 1; done
PTMI_LIB=$ab/libptmi_nopf.so run cfg1_plain_one --config 1 --overlap 0 && run cfg1_prefetch_one --config 1 --overlap 0
PTMI_LIB=$ab/libptmi_nopf.so run cfg3_plain --config 3 && run cfg3_prefetch --config 3
PTMI_LIB=$ab/libptmi_nopf.so run cfg2_plain --config 2 && run cfg2_prefetch --config 2
