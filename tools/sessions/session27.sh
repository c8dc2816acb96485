#!/bin/bash
# GPU session 27: grid of the grid-stride shade kernel (256-thread workgroups per CU)
set -o pipefail
out=gpurun_out/s27; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2; do
run cfg1_sh8_$i --config 1
for n in 5 10 16 32; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_sh$n.so run cfg1_sh${n}_$i --config 1 || exit 1; done
done
