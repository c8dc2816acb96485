#!/bin/bash
# GPU session 47: a sample (2.7e11 operand pairs) of short quotients against the IEEE expansion
set -o pipefail
mkdir -p gpurun_out/s47
timeout -k 10 500 ./tools/ubench/exact_div.bin > gpurun_out/s47/exact_div.log 2>&1; rc=$?; cat gpurun_out/s47/exact_div.log; exit $rc
