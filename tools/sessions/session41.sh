#!/bin/bash
# GPU session 41: shade reading the material field by field (86 VGPRs; with the 6-waves attribute 80 + 3 spilled dwords) against the
# whole-struct copy (94 VGPRs): parity subset first, then A/B
set -o pipefail
out=gpurun_out/s41; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_atlas_host.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_old.so run cfg1_old_$i --config 1 && run cfg1_lazy_$i --config 1 && PTMI_LIB=$ab/libptmi_w6.so run cfg1_lazy6_$i --config 1 || exit 1; done
PTMI_LIB=$ab/libptmi_old.so run cfg1_old_one --config 1 --overlap 0 && run cfg1_lazy_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_w6.so run cfg1_lazy6_one --config 1 --overlap 0
PTMI_LIB=$ab/libptmi_old.so run cfg2_old --config 2 && run cfg2_lazy --config 2 && PTMI_LIB=$ab/libptmi_w6.so run cfg2_lazy6 --config 2
PTMI_LIB=$ab/libptmi_old.so run cfg3_old --config 3 && run cfg3_lazy --config 3 && PTMI_LIB=$ab/libptmi_w6.so run cfg3_lazy6 --config 3
