#!/bin/bash
# GPU session 11: refill threshold of the global traversal kernels; config 2 with the exact image for its shadow rays
set -o pipefail
out=gpurun_out/s11; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2; do
run cfg3_g44_$i --config 3
for v in g52 g58; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_$v.so run cfg3_${v}_$i --config 3 || exit 1; done
done
run cfg2_g44 --config 2 --steps 4
for v in g52 g58; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_$v.so run cfg2_$v --config 2 --steps 4 || exit 1; done
run cfg2_g44b --config 2 --steps 4
