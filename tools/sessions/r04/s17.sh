#!/bin/bash
# round 4, GPU session 17: the from-memory traversal kernels deal the queue out to the eight XCDs in blocks (every XCD's L2 sees its own
# part of the picture) against group by group (-DPT_XCD_BLOCKS=0); parity first
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s17; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2 3; do
  TAG="cfg3 blocks per XCD" b --config 3 --steps 2
  TAG="cfg3 group by group" PTMI_LIB=$ab/libptmi_noxcd.so b --config 3 --steps 2
  TAG="cfg3 blocks per XCD, one stream" b --config 3 --steps 2 --overlap 0
  TAG="cfg3 group by group, one stream" PTMI_LIB=$ab/libptmi_noxcd.so b --config 3 --steps 2 --overlap 0
done 2>&1 | tee $out/ab_cfg3.txt
for round in 1 2; do
  TAG="cfg2 from memory, blocks per XCD" PTMI_OWN_EXTEND=8 PTMI_OWN_SHADOW=8 b --config 2 --steps 1
  TAG="cfg2 from memory, group by group" PTMI_OWN_EXTEND=8 PTMI_OWN_SHADOW=8 PTMI_LIB=$ab/libptmi_noxcd.so b --config 2 --steps 1
done 2>&1 | tee $out/ab_cfg2_memory.txt
