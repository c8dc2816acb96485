#!/bin/bash
# GPU session 25: compaction tile size (256 / 512 / 1024 ballot words per workgroup)
set -o pipefail
out=gpurun_out/s25; mkdir -p $out
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_t1024.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest_t1024.log 2>&1; rc=$?; tail -3 $out/pytest_t1024.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do
run cfg1_t256_$i --config 1 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_t512.so run cfg1_t512_$i --config 1 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_t1024.so run cfg1_t1024_$i --config 1 || exit 1
done
