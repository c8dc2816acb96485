#!/bin/bash
# GPU session 10: scheduling constants of the traversal loop on the 1 M-triangle scene (and Cornell, to see which are scene-specific)
set -o pipefail
out=gpurun_out/s10; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for c in 3 1; do
run cfg${c}_base --config $c
for v in r28 r44 n4 n12 l2 l6; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_$v.so run cfg${c}_$v --config $c || exit 1; done
run cfg${c}_base2 --config $c
done
