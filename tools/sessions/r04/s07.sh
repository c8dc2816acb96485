#!/bin/bash
# round 4, GPU session 7: how long the box stream keeps going after a vote (NODE_KEEP, NODE_STEPS) in the own-leaf kernels, config 1
# overlapped and one stream, config 2
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s07; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2; do
  TAG="base      " b
  for v in nk4 nk4s4 nk5 nk6 nk8 nk64 nk6ns12 nk64ns12; do TAG="$v       " PTMI_LIB=$ab/libptmi_$v.so b; done
done 2>&1 | tee $out/ab_cfg1.txt
echo "--- one stream"
for v in base nk4 nk4s4 nk5 nk6 nk8 nk64 nk6ns12 nk64ns12; do lib=$ab/libptmi_$v.so; [ $v = base ] && lib=$root/wgpu-path-tracing_amd/lib/libptmi.so; TAG="one stream $v" PTMI_LIB=$lib b --overlap 0; done 2>&1 | tee $out/ab_cfg1_one_stream.txt
echo "--- config 2 / 3"
for v in base nk4 nk6 nk64; do lib=$ab/libptmi_$v.so; [ $v = base ] && lib=$root/wgpu-path-tracing_amd/lib/libptmi.so; TAG="cfg2 $v" PTMI_LIB=$lib b --config 2 --steps 2; TAG="cfg3 $v" PTMI_LIB=$lib b --config 3 --steps 2; done 2>&1 | tee $out/ab_cfg23.txt
