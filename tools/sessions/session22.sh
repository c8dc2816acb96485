#!/bin/bash
# GPU session 22: tree rotations after the SAH build (depth-budgeted / unbounded / off): parity, then A/B
set -o pipefail
out=gpurun_out/s22; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_norot.so run cfg1_norot_$i --config 1 &&
run cfg1_rot14_$i --config 1 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_rot60.so run cfg1_rot60_$i --config 1 || exit 1
done
for i in 1 2; do
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_norot.so run cfg2_norot_$i --config 2 --steps 2 &&
run cfg2_rot_$i --config 2 --steps 2 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_rot60.so run cfg2_rot60_$i --config 2 --steps 2 || exit 1
done
