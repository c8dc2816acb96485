#!/bin/bash
# GPU session 24: long repeatability soak of the two-stream pipeline (100 rounds x 5 cases), fuzz once more on the final build
set -o pipefail
out=gpurun_out/s24; mkdir -p $out
timeout -k 10 500 python tools/soak_gpu.py 100 > $out/soak.log 2>&1; rc=$?; tail -3 $out/soak.log; [ $rc = 0 ] || exit $rc
timeout -k 10 500 python tools/fuzz_gpu.py 30 512 384 8 > $out/fuzz.log 2>&1; rc=$?; tail -2 $out/fuzz.log; exit $rc
