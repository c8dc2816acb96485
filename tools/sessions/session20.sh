#!/bin/bash
# GPU session 20: fuzz and soak of the final build (random scenes in every traversal / stream mode, extreme scales, determinism)
set -o pipefail
out=gpurun_out/s20; mkdir -p $out
timeout -k 10 500 python tools/fuzz_gpu.py 20 512 384 8 > $out/fuzz.log 2>&1; rc=$?; tail -3 $out/fuzz.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_scale_gpu.py > $out/fuzz_scale.log 2>&1; rc=$?; tail -4 $out/fuzz_scale.log; [ $rc = 0 ] || exit $rc
timeout -k 10 400 python tools/fuzz_big_gpu.py > $out/fuzz_big.log 2>&1; rc=$?; tail -3 $out/fuzz_big.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/soak_gpu.py 10 > $out/soak.log 2>&1; rc=$?; tail -3 $out/soak.log; exit $rc
