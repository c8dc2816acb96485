#!/bin/bash
# GPU session 30: what would fewer VALU instructions in `shade` buy now that the shadow kernel runs beside it?
# perf_mode = 1 (fast rcp / sqrt: 39 % fewer VALU instructions, not bit-exact) as the upper bound, three interleaved rounds
set -o pipefail
out=gpurun_out/s30; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run cfg1_exact_$i --config 1 && run cfg1_perf_$i --config 1 --perf-mode 1 || exit 1; done
run cfg1_exact_one --config 1 --overlap 0 && run cfg1_perf_one --config 1 --perf-mode 1 --overlap 0
run cfg3_exact --config 3 && run cfg3_perf --config 3 --perf-mode 1
