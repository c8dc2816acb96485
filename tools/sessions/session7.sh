#!/bin/bash
# GPU session 7: 8-byte hit records + packed ray state: parity, then A/B against the previous build
set -o pipefail
out=gpurun_out/s7; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log; [ $rc = 0 ] || exit $rc
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_qc128.so wgpu-path-tracing_amd/lib/libptmi.so 3 --config 1 > $out/ab_cfg1.log 2>&1; cat $out/ab_cfg1.log
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_qc128.so wgpu-path-tracing_amd/lib/libptmi.so 1 --config 2 --steps 4 > $out/ab_cfg2.log 2>&1; cat $out/ab_cfg2.log
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_qc128.so wgpu-path-tracing_amd/lib/libptmi.so 2 --config 3 > $out/ab_cfg3.log 2>&1; cat $out/ab_cfg3.log
