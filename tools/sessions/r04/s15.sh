#!/bin/bash
# round 4, GPU session 15: sample_bsdf with the lobes' common prefix (two draws, sin / cos, the frame around the normal) and the GGX
# half-vector executed once per wave instead of once per lobe — the whole GPU suite, then same-box A/B against the previous build
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s15; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_own_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_own_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_own_leaves.log | head -100; exit 1; }
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
prev=$root/wgpu-path-tracing_amd/lib/ab/libptmi_prev.so
for round in 1 2 3; do
  TAG="cfg1 merged lobes" b
  TAG="cfg1 previous    " PTMI_LIB=$prev b
  TAG="cfg1 merged, one stream  " b --overlap 0
  TAG="cfg1 previous, one stream" PTMI_LIB=$prev b --overlap 0
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 merged lobes" b --config 2 --steps 2
  TAG="cfg2 previous    " PTMI_LIB=$prev b --config 2 --steps 2
  TAG="cfg3 merged lobes" b --config 3 --steps 2
  TAG="cfg3 previous    " PTMI_LIB=$prev b --config 3 --steps 2
done 2>&1 | tee $out/ab_cfg23.txt
