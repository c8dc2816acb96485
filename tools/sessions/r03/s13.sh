#!/bin/bash
# round 3, GPU session 13: unroll count of the box-step loop in the spilling traversal variants (configs 3 and 2 walk them): the default
# (8 copies, twice) against 4 / 2 / 1 — the quantised closest-hit kernel is 7 157 instructions (~46 KB) at 8. Parity of the smallest first.
set -o pipefail
out=gpurun_out/r03_s13; mkdir -p $out
ab=$PWD/wgpu-path-tracing_amd/lib/ab
PTMI_LIB=$ab/libptmi_unroll1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extend_parity or deep_tree or irregular or occluded" > $out/pytest_unroll1.log 2>&1; rc=$?; tail -3 $out/pytest_unroll1.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so
for i in 1 2 3; do
  run c3_u8_$i $def --config 3 && run c3_u4_$i $ab/libptmi_unroll4.so --config 3 && run c3_u2_$i $ab/libptmi_unroll2.so --config 3 && run c3_u1_$i $ab/libptmi_unroll1.so --config 3 || exit 1
done
for i in 1 2; do
  run c2_u8_$i $def --config 2 --steps 2 && run c2_u2_$i $ab/libptmi_unroll2.so --config 2 --steps 2 && run c2_u1_$i $ab/libptmi_unroll1.so --config 2 --steps 2 || exit 1
done
run c3_u8_one $def --config 3 --overlap 0 && run c3_u1_one $ab/libptmi_unroll1.so --config 3 --overlap 0
