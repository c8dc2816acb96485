#!/bin/bash
# GPU session 28: shade grid 8 / 16 / 32 / 64 workgroups per CU, five interleaved rounds; config 3 once
set -o pipefail
out=gpurun_out/s28; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0']['shade'], d['gpu_ms_rank0'])"; }
for i in 1 2 3 4 5; do
run cfg1_sh8_$i --config 1
for n in 16 32 64; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_sh$n.so run cfg1_sh${n}_$i --config 1 || exit 1; done
done
run cfg3_sh8 --config 3; for n in 16 32 64; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_sh$n.so run cfg3_sh${n} --config 3 || exit 1; done
run cfg3_sh8b --config 3; for n in 16 32 64; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_sh$n.so run cfg3_sh${n}b --config 3 || exit 1; done
