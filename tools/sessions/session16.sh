#!/bin/bash
# GPU session 16: larger wavefront batches (64 frames = 133 M paths at 1080p) against the default 32
set -o pipefail
out=gpurun_out/s16; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config']['frames_per_batch'], d['segments_by_bounce_rank0'])"; }
for i in 1 2; do
run cfg1_f32_$i --config 1 &&
run cfg1_f64_$i --config 1 --frames-per-step 64 --frames-per-batch 64 --steps 1 &&
run cfg1_f128_$i --config 1 --frames-per-step 128 --frames-per-batch 128 --steps 1 || exit 1
done
run cfg3_f32 --config 3 && run cfg3_f64 --config 3 --frames-per-step 64 --frames-per-batch 64 --steps 1
