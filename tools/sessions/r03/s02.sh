#!/bin/bash
# round 3, GPU session 2: the multi-device handle (N = 1 through RCCL, N = 2, 3, 8, 5 on one device with loopback copies), then
# where the work-list build's lanes and iterations go (diagnostic build, tools/lane_stats.py)
set -o pipefail
out=gpurun_out/r03_s02; mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $out/pytest_multi.log 2>&1; rc=$?; tail -15 $out/pytest_multi.log
ab=$PWD/wgpu-path-tracing_amd/lib/ab
PTMI_LIB=$ab/libptmi_util.so PTMI_OPTS='{"traversal": 2, "worklist": 1}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_lds_off.json 2> $out/lane_off.err || tail -3 $out/lane_off.err
PTMI_LIB=$ab/libptmi_util.so PTMI_OPTS='{"traversal": 2, "worklist": 2}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_lds_wl.json 2> $out/lane_wl.err || tail -3 $out/lane_wl.err
cat $out/lane_lds_off.json $out/lane_lds_wl.json
exit $rc
