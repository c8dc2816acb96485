#!/bin/bash
# round 3, GPU session 1: the per-wave work list (ptmi_options.worklist = 2) — parity first, then same-box A/B against the per-lane
# loop with both traversal kernels in their full LDS variant (--traversal lds), and the default build beside it
set -o pipefail
out=gpurun_out/r03_s01; mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_worklist.py -m gpu -x -q > $out/pytest_wl.log 2>&1; rc=$?; tail -5 $out/pytest_wl.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('worklist_used'))"; }
for i in 1 2 3; do
  run lds_off_$i --config 1 --traversal lds --worklist 1 && run lds_wl_$i --config 1 --traversal lds --worklist 2 && run auto_$i --config 1 || exit 1
done
run lds_off_one --config 1 --traversal lds --worklist 1 --overlap 0 && run lds_wl_one --config 1 --traversal lds --worklist 2 --overlap 0 && run auto_one --config 1 --overlap 0 || exit 1
