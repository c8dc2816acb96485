#!/bin/bash
# GPU session 35: which of the two 12-byte records costs config 3 its 4 %? old = both 16, f3 = both 12, l16 = L 16 / SC 12, sc16 = L 12 / SC 16
set -o pipefail
out=gpurun_out/s35; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg3_old_$i --config 3 && run cfg3_f3_$i --config 3 && PTMI_LIB=$ab/libptmi_l16.so run cfg3_l16_$i --config 3 && PTMI_LIB=$ab/libptmi_sc16.so run cfg3_sc16_$i --config 3 || exit 1; done
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg1_old_$i --config 1 && run cfg1_f3_$i --config 1 && PTMI_LIB=$ab/libptmi_l16.so run cfg1_l16_$i --config 1 && PTMI_LIB=$ab/libptmi_sc16.so run cfg1_sc16_$i --config 1 || exit 1; done
