#!/bin/bash
# GPU session 29: priority of the shadow stream (lowest, the default, against normal), five interleaved rounds
set -o pipefail
out=gpurun_out/s29; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3 4 5; do run cfg1_low_$i --config 1 && PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_prio.so run cfg1_normal_$i --config 1 || exit 1; done
for i in 1 2; do run cfg3_low_$i --config 3 && PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_prio.so run cfg3_normal_$i --config 3 || exit 1; done
