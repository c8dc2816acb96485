#!/bin/bash
# round 3, GPU session 16: fuzz and soak of the final build — 24 random scenes x 8 ways of computing the same image (tree as uploaded /
# rebuilt, LDS / quantised / exact image, one stream / shadow stream / two lanes, state in place / following the queue, serial tail,
# work list), scaled scenes, big scenes, repeatability — every float against the oracle
set -o pipefail
out=gpurun_out/r03_s16; mkdir -p $out
timeout -k 10 900 python tools/fuzz_gpu.py 24 384 256 6 > $out/fuzz.log 2>&1; rc=$?; tail -3 $out/fuzz.log; [ $rc = 0 ] || exit $rc
timeout -k 10 600 python tools/fuzz_scale_gpu.py > $out/fuzz_scale.log 2>&1; rc=$?; tail -3 $out/fuzz_scale.log; [ $rc = 0 ] || exit $rc
timeout -k 10 900 python tools/fuzz_big_gpu.py > $out/fuzz_big.log 2>&1; rc=$?; tail -3 $out/fuzz_big.log; [ $rc = 0 ] || exit $rc
timeout -k 10 600 python tools/soak_gpu.py 10 > $out/soak.log 2>&1; rc=$?; tail -3 $out/soak.log; exit $rc
