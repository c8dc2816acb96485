#!/bin/bash
# GPU session 52: overlap 5 (whole batches alternate between two lanes; the next starts when this one has compacted its last bounce
# but one) — parity of the stream modes, then A/B over several steps (the overlap is ACROSS steps) against overlap 1
set -o pipefail
out=gpurun_out/s52; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped or two_lanes" > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
for i in 1 2 3; do run cfg1_o1_$i --config 1 --steps 8 --warmup 2 --overlap 1 && run cfg1_o5_$i --config 1 --steps 8 --warmup 2 --overlap 5 || exit 1; done
run cfg3_o1 --config 3 --steps 6 --warmup 2 --overlap 1 && run cfg3_o5 --config 3 --steps 6 --warmup 2 --overlap 5
run cfg2_o1 --config 2 --overlap 1 && run cfg2_o5 --config 2 --overlap 5
run cfg4_o1 --config 4 --overlap 1 && run cfg4_o5 --config 4 --overlap 5
