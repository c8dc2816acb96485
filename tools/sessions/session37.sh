#!/bin/bash
# GPU session 37: radiance stride per dispatch with whole 16-byte accesses at stride 4 — parity subset, then A/B
set -o pipefail
out=gpurun_out/s37; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg3_old_$i --config 3 && run cfg3_auto_$i --config 3 || exit 1; done
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg1_old_$i --config 1 && run cfg1_auto_$i --config 1 && PTMI_LIB=$ab/libptmi_ls4.so run cfg1_ls4_$i --config 1 || exit 1; done
for i in 1 2; do PTMI_LIB=$ab/libptmi_old.so run cfg2_old_$i --config 2 && run cfg2_auto_$i --config 2 && PTMI_LIB=$ab/libptmi_ls4.so run cfg2_ls4_$i --config 2 || exit 1; done
