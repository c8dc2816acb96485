#!/bin/bash
# GPU session 45: shadow kernel with one record allocation (80 scalar registers in the node-cache variant) — GPU suite, then the
# any-hit kernel from the node cache (two workgroups per CU, triangles through L1/L2) against the full LDS image (one workgroup)
set -o pipefail
out=gpurun_out/s45; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do run cfg1_full_$i --config 1 && PTMI_LIB=$ab/libptmi_snc.so run cfg1_nodecache_$i --config 1 || exit 1; done
run cfg1_full_one --config 1 --overlap 0 && PTMI_LIB=$ab/libptmi_snc.so run cfg1_nodecache_one --config 1 --overlap 0
run cfg4_full --config 4 && PTMI_LIB=$ab/libptmi_snc.so run cfg4_nodecache --config 4
