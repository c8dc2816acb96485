#!/bin/bash
# GPU session 39: soak and fuzz of the build with 44-byte records and the per-dispatch radiance stride
set -o pipefail
out=gpurun_out/s39; mkdir -p $out
timeout -k 10 400 python tools/soak_gpu.py 40 > $out/soak.log 2>&1; rc=$?; tail -3 $out/soak.log; [ $rc = 0 ] || exit $rc
timeout -k 10 500 python tools/fuzz_gpu.py 30 512 384 8 > $out/fuzz.log 2>&1; rc=$?; tail -2 $out/fuzz.log; [ $rc = 0 ] || exit $rc
timeout -k 10 400 python tools/fuzz_scale_gpu.py > $out/fuzz_scale.log 2>&1; rc=$?; tail -2 $out/fuzz_scale.log; exit $rc
