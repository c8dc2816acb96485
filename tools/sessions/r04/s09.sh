#!/bin/bash
# round 4, GPU session 9: exact nodes with 16-bit references and stack entries at two workgroups per CU (the default for scenes of up
# to 4 096 triangles whose nodes fit) against the quantised nodes it replaces, config 1
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s09; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2 3; do
  TAG="16-bit both     " b
  TAG="quantised both  " PTMI_OWN_EXTEND=17 PTMI_OWN_SHADOW=17 b
  TAG="16-bit extend   " PTMI_OWN_SHADOW=17 b
  TAG="16-bit shadow   " PTMI_OWN_EXTEND=17 b
  TAG="leaves 1        " b --leaves 1
done 2>&1 | tee $out/ab_cfg1.txt
TAG="one stream 16-bit   " b --overlap 0
TAG="one stream quantised" PTMI_OWN_EXTEND=17 PTMI_OWN_SHADOW=17 b --overlap 0
TAG="cfg0" b --config 0
TAG="cfg4" b --config 4 --steps 2
