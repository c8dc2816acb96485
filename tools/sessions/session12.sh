#!/bin/bash
# GPU session 12: non-temporal hints on the one-touch streams (A/B), parity of that build
set -o pipefail
out=gpurun_out/s12; mkdir -p $out
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_nt.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/pytest_nt.log 2>&1; rc=$?; tail -3 $out/pytest_nt.log; [ $rc = 0 ] || exit $rc
for c in 1 3 2; do extra=""; [ $c = 2 ] && extra="--steps 4"
bash tools/ab.sh wgpu-path-tracing_amd/lib/libptmi.so wgpu-path-tracing_amd/lib/ab/libptmi_nt.so 2 --config $c $extra > $out/ab_cfg$c.log 2>&1; cat $out/ab_cfg$c.log; done
