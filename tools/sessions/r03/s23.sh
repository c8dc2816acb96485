#!/bin/bash
# round 3, GPU session 23: the traversal hierarchy built on the GPU (ptmi_options.tree_builder = 2): parity, upload times, and what the
# Morton-order tree costs in traversal (interleaved A/B on configs 3, 2, 1)
set -o pipefail
out=gpurun_out/r03_s23; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_tree_builder.py -m gpu -x -q > $out/pytest_tb.log 2>&1; rc=$?; tail -5 $out/pytest_tb.log; [ $rc = 0 ] || exit $rc
python tools/time_upload.py > $out/upload.log 2>&1; cat $out/upload.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['upload_ms_rank0'])"; }
for i in 1 2; do
  run c3_host_$i --config 3 --tree-builder 1 && run c3_gpu_$i --config 3 --tree-builder 2 || exit 1
  run c1_host_$i --config 1 --tree-builder 1 && run c1_gpu_$i --config 1 --tree-builder 2 || exit 1
  run c2_host_$i --config 2 --steps 2 --tree-builder 1 && run c2_gpu_$i --config 2 --steps 2 --tree-builder 2 || exit 1
done
