#!/bin/bash
# GPU session 40: shade at 6 waves per SIMD (80 VGPRs + 12 spilled dwords) against 5 (94 VGPRs), now that it runs beside the shadow kernel
set -o pipefail
out=gpurun_out/s40; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do run cfg1_w5_$i --config 1 && PTMI_LIB=$ab/libptmi_w6.so run cfg1_w6_$i --config 1 || exit 1; done
run cfg1_w5_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_w6.so run cfg1_w6_one --config 1 --overlap 0
run cfg3_w5 --config 3 && PTMI_LIB=$ab/libptmi_w6.so run cfg3_w6 --config 3
