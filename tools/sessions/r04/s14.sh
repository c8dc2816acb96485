#!/bin/bash
# round 4, GPU session 14: the whole GPU suite in both leaf modes on the sources with the two-workgroup variant of configs[2], then
# larger own leaves for the scene that walks memory (configs[3]: fewer node fetches against more triangle tests)
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s14; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_own_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_own_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_own_leaves.log | head -100; exit 1; }
PTMI_TEST_LEAVES=1 timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_reference_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_reference_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_reference_leaves.log | head -100; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" && echo smoke ok
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2; do
  for k in 2 3 4 6 8; do TAG="cfg3 leaf_tris $k" b --config 3 --steps 2 --leaf-tris $k; done
done 2>&1 | tee $out/ab_cfg3_leaf_tris.txt
