#!/bin/bash
# round 4, GPU session 6: scheduling constants of the own-leaf kernels (A/B builds) and the shape of the own tree (leaf_tris, the collapse
# costs), config 1 overlapped and on one stream; the tree parameters on config 2 as well
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s06; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2; do
  TAG="base      " b
  for v in ns12 ns6 ls2 ls8 nk2 nk4 lk2; do TAG="$v       " PTMI_LIB=$ab/libptmi_$v.so b; done
  TAG="leaf_tris 1" b --leaf-tris 1
  TAG="leaf_tris 4" b --leaf-tris 4
  TAG="c_tri 0.6  " PTMI_OWN_C_TRI=0.6 b --leaf-tris 4
  TAG="c_tri 1.3  " PTMI_OWN_C_TRI=1.3 b
  TAG="c_open 0   " PTMI_OWN_C_OPEN=0 b
  TAG="c_open 0.8 " PTMI_OWN_C_OPEN=0.8 b --leaf-tris 4
done 2>&1 | tee $out/ab_cfg1.txt
echo "--- one stream"
for v in base ns12 ns6 ls2 ls8 nk2 nk4 lk2; do lib=$ab/libptmi_$v.so; [ $v = base ] && lib=$root/wgpu-path-tracing_amd/lib/libptmi.so; TAG="one stream $v" PTMI_LIB=$lib b --overlap 0; done 2>&1 | tee $out/ab_cfg1_one_stream.txt
echo "--- config 2"
for round in 1 2; do
  TAG="cfg2 base      " b --config 2 --steps 2
  TAG="cfg2 leaf_tris 4" b --config 2 --steps 2 --leaf-tris 4
  TAG="cfg2 c_tri 0.6  " PTMI_OWN_C_TRI=0.6 b --config 2 --steps 2 --leaf-tris 4
  TAG="cfg2 c_open 0   " PTMI_OWN_C_OPEN=0 b --config 2 --steps 2
done 2>&1 | tee $out/ab_cfg2.txt
