#!/bin/bash
# GPU session 13: both compactions of a bounce in one launch: whole suite, then A/B
set -o pipefail
out=gpurun_out/s13; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
for c in 1 3; do
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_nos2.so wgpu-path-tracing_amd/lib/libptmi.so 3 --config $c > $out/ab_cfg$c.log 2>&1; cat $out/ab_cfg$c.log; done
