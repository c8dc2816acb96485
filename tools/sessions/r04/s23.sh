#!/bin/bash
# round 4, GPU session 23: the ray's three reciprocals without range tests of their own (a regular ray's direction lies where the short form is exact)
# parity, then A/B against the previous build
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s23; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
prev=$root/wgpu-path-tracing_amd/lib/ab/libptmi_prev.so
for round in 1 2 3; do
  TAG="cfg1 new     " b
  TAG="cfg1 previous" PTMI_LIB=$prev b
  TAG="cfg1 new, one stream     " b --overlap 0
  TAG="cfg1 previous, one stream" PTMI_LIB=$prev b --overlap 0
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 new     " b --config 2 --steps 2
  TAG="cfg2 previous" PTMI_LIB=$prev b --config 2 --steps 2
  TAG="cfg3 new     " b --config 3 --steps 2
  TAG="cfg3 previous" PTMI_LIB=$prev b --config 3 --steps 2
done 2>&1 | tee $out/ab_cfg23.txt
