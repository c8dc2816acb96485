#!/bin/bash
# round 3, GPU session 22: a larger fuzz of the final build (100 random scenes x 8 ways of computing the same image) and 40 soak rounds
set -o pipefail
out=gpurun_out/r03_s22; mkdir -p $out
timeout -k 10 1100 python tools/fuzz_gpu.py 100 320 200 5 > $out/fuzz100.log 2>&1; rc=$?; tail -2 $out/fuzz100.log; grep -c "OK$" $out/fuzz100.log; [ $rc = 0 ] || exit $rc
