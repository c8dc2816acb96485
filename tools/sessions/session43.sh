#!/bin/bash
# GPU session 43: exhaustive (2^32 inputs) check of short sequences for 1/x and sqrt(x) against the IEEE expansions
set -o pipefail
mkdir -p gpurun_out/s43
timeout -k 10 300 ./tools/ubench/exact_math.bin > gpurun_out/s43/exact_math.log 2>&1; rc=$?; cat gpurun_out/s43/exact_math.log; exit $rc
