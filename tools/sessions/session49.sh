#!/bin/bash
# GPU session 49: refill threshold of the LDS traversal kernels again, now that a refill's three reciprocals are cheaper
set -o pipefail
out=gpurun_out/s49; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do run cfg1_r36_$i --config 1 && PTMI_LIB=$ab/libptmi_r44.so run cfg1_r44_$i --config 1 && PTMI_LIB=$ab/libptmi_r52.so run cfg1_r52_$i --config 1 && PTMI_LIB=$ab/libptmi_r28.so run cfg1_r28_$i --config 1 || exit 1; done
run cfg1_r36_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_r44.so run cfg1_r44_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_r52.so run cfg1_r52_one --config 1 --overlap 0
