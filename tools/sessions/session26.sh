#!/bin/bash
# GPU session 26: what the library's per-kernel HIP events cost (bench.py --timing 3 / 1 / 0), parity on the 1024-word tiles
set -o pipefail
out=gpurun_out/s26; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run cfg1_t3_$i --config 1 --timing 3 && run cfg1_t1_$i --config 1 --timing 1 && run cfg1_t0_$i --config 1 --timing 0 || exit 1; done
run cfg3_t3 --config 3 --timing 3 && run cfg3_t0 --config 3 --timing 0
