#!/bin/bash
# round 4, GPU session 29: two old switches on the final kernels — non-temporal loads / stores for the ray-state streams (-DPT_NT=1) and the
# radiance at a 16-byte stride (-DPT_L_STRIDE=4: no 12-byte record straddles a 32-byte sector) — same box, interleaved
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s29; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], 'extend %.2f shade %.2f shadow %.2f raygen %.2f accumulate %.2f' % (k['extend'], k['shade'], k['shadow'], k['raygen'], k['accumulate']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2 3; do
  TAG="cfg1 base            " b
  TAG="cfg1 non-temporal    " PTMI_LIB=$ab/libptmi_nt.so b
  TAG="cfg1 radiance stride 4" PTMI_LIB=$ab/libptmi_l4.so b
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 base            " b --config 2 --steps 2
  TAG="cfg2 non-temporal    " PTMI_LIB=$ab/libptmi_nt.so b --config 2 --steps 2
  TAG="cfg2 radiance stride 4" PTMI_LIB=$ab/libptmi_l4.so b --config 2 --steps 2
  TAG="cfg3 base (stride 4) " b --config 3 --steps 2
  TAG="cfg3 non-temporal    " PTMI_LIB=$ab/libptmi_nt.so b --config 3 --steps 2
done 2>&1 | tee $out/ab_cfg23.txt
