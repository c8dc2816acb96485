#!/bin/bash
# round 3, GPU session 10: the whole GPU suite after the pt.wgsl:647 change (shade's miss branch reads the throughput), and what it costs
set -o pipefail
out=gpurun_out/r03_s10; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -6 $out/pytest_gpu.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run c1_$i --config 1 || exit 1; done
run c3_1 --config 3 && run c2_1 --config 2 && run c4_1 --config 4
