#!/bin/bash
# round 4, GPU session 11: cornell_spheres (configs[2]) from two workgroups per CU — quantised nodes with 16-bit references, 8 16-bit
# entries per lane, the node stack spills (PT_VARIANT_OWN_QLDS16_NODES) — against the one-workgroup kernels it ran as so far
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s11; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_own_leaves.py -m gpu -x -q > $out/pytest_own.log 2>&1 || { tail -30 $out/pytest_own.log; exit 1; }
tail -2 $out/pytest_own.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2; do
  TAG="cfg2 new pick            " b --config 2 --steps 2
  TAG="cfg2 one workgroup (r04) " PTMI_OWN_EXTEND=7 PTMI_OWN_SHADOW=7 b --config 2 --steps 2
  TAG="cfg2 extend 2 / shadow 1 " PTMI_OWN_SHADOW=7 b --config 2 --steps 2
  TAG="cfg2 extend 1 / shadow 2 " PTMI_OWN_EXTEND=7 b --config 2 --steps 2
  TAG="cfg2 leaf_tris 3 (11 entries)" b --config 2 --steps 2 --leaf-tris 3
  TAG="cfg2 leaf_tris 4 (12 entries)" b --config 2 --steps 2 --leaf-tris 4
done 2>&1 | tee $out/ab_cfg2.txt
for round in 1 2; do
  TAG="cfg1 default (exact nodes, 16-bit references)" b
  TAG="cfg1 quantised, 16-bit references, 15 entries" PTMI_OWN_EXTEND=21 PTMI_OWN_SHADOW=21 b
done 2>&1 | tee $out/ab_cfg1.txt
