#!/bin/bash
# round 3, GPU session 21: chunks of the ray queue CLAIMED by the waves (-DPT_DYNAMIC_CLAIM=1) instead of a fixed share per wave — so
# that a traversal grid that is not fully resident at launch does not run a second round. Parity of the variant, then: does it cost
# anything by itself, and do the overlaps that lost with fixed shares (pipeline = 2, overlap = 3) win with it?
set -o pipefail
out=gpurun_out/r03_s21; mkdir -p $out
ab=$PWD/wgpu-path-tracing_amd/lib/ab; dyn=$ab/libptmi_dyn.so; def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so
PTMI_LIB=$dyn timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest_dyn.log 2>&1; rc=$?; tail -3 $out/pytest_dyn.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do
  run c1_def_$i $def --config 1 --steps 8 && run c1_dyn_$i $dyn --config 1 --steps 8 && run c1_dyn_pipe_$i $dyn --config 1 --steps 8 --pipeline 2 && run c1_dyn_ov3_$i $dyn --config 1 --steps 8 --overlap 3 && run c1_def_ov3_$i $def --config 1 --steps 8 --overlap 3 || exit 1
done
for i in 1 2; do
  run c3_def_$i $def --config 3 --steps 4 && run c3_dyn_$i $dyn --config 3 --steps 4 && run c3_dyn_pipe_$i $dyn --config 3 --steps 4 --pipeline 2 || exit 1
  run c2_def_$i $def --config 2 && run c2_dyn_$i $dyn --config 2 && run c2_dyn_pipe_$i $dyn --config 2 --pipeline 2 || exit 1
done
run c1_def_one $def --config 1 --overlap 0 && run c1_dyn_one $dyn --config 1 --overlap 0
