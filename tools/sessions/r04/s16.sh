#!/bin/bash
# round 4, GPU session 16: shade at 6 / 7 waves per SIMD now that it needs 88 registers (80 + 3 spilled dwords at 6 waves; round 3: 80 + 12)
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s16; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2 3; do
  TAG="cfg1 5 waves (88 registers)" b
  TAG="cfg1 6 waves" PTMI_LIB=$ab/libptmi_sw6.so b
  TAG="cfg1 7 waves" PTMI_LIB=$ab/libptmi_sw7.so b
  TAG="cfg1 5 waves, one stream" b --overlap 0
  TAG="cfg1 6 waves, one stream" PTMI_LIB=$ab/libptmi_sw6.so b --overlap 0
  TAG="cfg1 7 waves, one stream" PTMI_LIB=$ab/libptmi_sw7.so b --overlap 0
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 5 waves" b --config 2 --steps 2
  TAG="cfg2 6 waves" PTMI_LIB=$ab/libptmi_sw6.so b --config 2 --steps 2
  TAG="cfg3 5 waves" b --config 3 --steps 2
  TAG="cfg3 6 waves" PTMI_LIB=$ab/libptmi_sw6.so b --config 3 --steps 2
done 2>&1 | tee $out/ab_cfg23.txt
