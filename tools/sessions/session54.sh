#!/bin/bash
# GPU session 54: the large fuzz (big random scenes through every traversal variant) and the extreme-scale fuzz on the final build
set -o pipefail
out=gpurun_out/s54; mkdir -p $out
timeout -k 10 900 python tools/fuzz_big_gpu.py > $out/fuzz_big.log 2>&1; rc=$?; tail -3 $out/fuzz_big.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/edge_more_gpu.py > $out/edge_more.log 2>&1; rc=$?; tail -3 $out/edge_more.log; exit $rc
