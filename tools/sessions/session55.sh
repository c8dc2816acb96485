#!/bin/bash
# GPU session 55: path ids by 16 x 4 pixel tiles instead of row by row — parity (whole suite), then A/B
set -o pipefail
out=gpurun_out/s55; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_linear.so run cfg1_rows_$i --config 1 && run cfg1_tiles_$i --config 1 || exit 1; done
PTMI_LIB=$ab/libptmi_linear.so run cfg1_rows_one --config 1 --overlap 0 && run cfg1_tiles_one --config 1 --overlap 0
for i in 1 2; do PTMI_LIB=$ab/libptmi_linear.so run cfg3_rows_$i --config 3 && run cfg3_tiles_$i --config 3 || exit 1; done
PTMI_LIB=$ab/libptmi_linear.so run cfg2_rows --config 2 && run cfg2_tiles --config 2
