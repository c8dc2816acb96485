#!/bin/bash
# round 4, GPU session 26: the any-hit kernel from ONE workgroup per CU beside the main stream (4 of 8 wave slots per SIMD, 236 of 512
# registers: three `shade` waves fit beside it on every SIMD; two workgroups fill the wave slots and the kernels take turns on a CU)
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s26; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2 3; do
  TAG="cfg1 shadow 2 workgroups per CU" b
  TAG="cfg1 shadow 1 workgroup per CU " PTMI_OWN_SHADOW=30 b
  TAG="cfg1 shadow 1, extend 1        " PTMI_OWN_SHADOW=30 PTMI_OWN_EXTEND=30 b
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg4 shadow 2 workgroups per CU" b --config 4 --steps 1
  TAG="cfg4 shadow 1 workgroup per CU " PTMI_OWN_SHADOW=30 b --config 4 --steps 1
done 2>&1 | tee $out/ab_cfg4.txt
