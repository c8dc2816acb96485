#!/bin/bash
# round 4, GPU session 24: the refill threshold of the LDS kernels again, now that a refill costs 35 vector instructions and seven loads less
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s24; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2 3; do
  TAG="cfg1 refill at 36" b --overlap 0
  TAG="cfg1 refill at 44" PTMI_LIB=$ab/libptmi_rf44.so b --overlap 0
  TAG="cfg1 refill at 52" PTMI_LIB=$ab/libptmi_rf52.so b --overlap 0
  TAG="cfg1 refill at 36, two streams" b
  TAG="cfg1 refill at 44, two streams" PTMI_LIB=$ab/libptmi_rf44.so b
  TAG="cfg1 refill at 52, two streams" PTMI_LIB=$ab/libptmi_rf52.so b
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 refill at 36" b --config 2 --steps 2
  TAG="cfg2 refill at 44" PTMI_LIB=$ab/libptmi_rf44.so b --config 2 --steps 2
  TAG="cfg2 refill at 52" PTMI_LIB=$ab/libptmi_rf52.so b --config 2 --steps 2
done 2>&1 | tee $out/ab_cfg2.txt
