#!/bin/bash
# GPU session 46: any-hit kernel from the node cache with ONE workgroup per CU (leaves half of every CU's LDS and wave slots to
# the main stream's kernels) against the full LDS image
set -o pipefail
out=gpurun_out/s46; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do run cfg1_full_$i --config 1 && PTMI_LIB=$ab/libptmi_snc1.so run cfg1_nc1_$i --config 1 || exit 1; done
run cfg1_full_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_snc1.so run cfg1_nc1_one --config 1 --overlap 0
