#!/bin/bash
# round 4, GPU session 12: the own triangle images laid out so that a leaf lies in one 128-byte line (ptmi_api.hip place_leaves)
# against the packed array (PTMI_TRI_PAD=0), configs 3, 1, 2
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s12; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py -m gpu -x -q > $out/pytest_own.log 2>&1 || { tail -30 $out/pytest_own.log; exit 1; }
tail -2 $out/pytest_own.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2 3; do
  TAG="cfg3 leaf in one line" b --config 3 --steps 2
  TAG="cfg3 packed          " PTMI_TRI_PAD=0 b --config 3 --steps 2
done 2>&1 | tee $out/ab_cfg3.txt
for round in 1 2; do
  TAG="cfg1 leaf in one line" b
  TAG="cfg1 packed          " PTMI_TRI_PAD=0 b
  TAG="cfg2 leaf in one line" b --config 2 --steps 2
  TAG="cfg2 packed          " PTMI_TRI_PAD=0 b --config 2 --steps 2
done 2>&1 | tee $out/ab_cfg12.txt
