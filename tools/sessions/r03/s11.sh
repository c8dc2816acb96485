#!/bin/bash
# round 3, GPU session 11: what the HIP events around every launch cost (--timing 3 vs 1), interleaved, config 1
set -o pipefail
out=gpurun_out/r03_s11; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['gpu_ms_rank0'])"; }
for i in 1 2 3 4; do
  run t3_$i --config 1 --timing 3 --steps 4 && run t1_$i --config 1 --timing 1 --steps 4 || exit 1
done
