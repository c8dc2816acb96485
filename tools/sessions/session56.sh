#!/bin/bash
# GPU session 56: shade requesting the whole triangle record before its first use (the short forms' range tests are branches; loads
# placed after one are not issued before it) — parity, then A/B
set -o pipefail
out=gpurun_out/s56; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_atlas_host.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg1_old_$i --config 1 && run cfg1_hoist_$i --config 1 || exit 1; done
PTMI_LIB=$ab/libptmi_old.so run cfg1_old_one --config 1 --overlap 0 && run cfg1_hoist_one --config 1 --overlap 0
PTMI_LIB=$ab/libptmi_old.so run cfg3_old --config 3 && run cfg3_hoist --config 3
PTMI_LIB=$ab/libptmi_old.so run cfg2_old --config 2 && run cfg2_hoist --config 2
