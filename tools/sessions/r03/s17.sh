#!/bin/bash
# round 3, GPU session 17: 8 waves per SIMD for the kernels that walk the scene from global memory (amdgpu_waves_per_eu(8): 58 VGPRs /
# 78 SGPRs instead of 74 / 102, no scratch; LDS top-of-tree cache 128 instead of 256 nodes so that 8 workgroups of 20 KB fit a CU)
# against the default's 6 — parity of the variant first, then interleaved A/B on configs 3 and 2
set -o pipefail
out=gpurun_out/r03_s17; mkdir -p $out
ab=$PWD/wgpu-path-tracing_amd/lib/ab
PTMI_LIB=$ab/libptmi_gw8.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extend_parity or deep_tree or irregular or occluded or fuzz" > $out/pytest_gw8.log 2>&1; rc=$?; tail -3 $out/pytest_gw8.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so
for i in 1 2 3; do
  run c3_def_$i $def --config 3 && run c3_gw8_$i $ab/libptmi_gw8.so --config 3 && run c3_gw8u2_$i $ab/libptmi_gw8u2.so --config 3 && run c3_gw7_$i $ab/libptmi_gw7.so --config 3 && run c3_q128_$i $ab/libptmi_q128.so --config 3 && run c3_gw8q256_$i $ab/libptmi_gw8q256.so --config 3 || exit 1
done
for i in 1 2; do
  run c2_def_$i $def --config 2 --steps 2 && run c2_gw8_$i $ab/libptmi_gw8.so --config 2 --steps 2 && run c2_gw7_$i $ab/libptmi_gw7.so --config 2 --steps 2 || exit 1
done
run c3_def_one $def --config 3 --overlap 0 && run c3_gw8_one $ab/libptmi_gw8.so --config 3 --overlap 0
