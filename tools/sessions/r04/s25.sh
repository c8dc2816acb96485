#!/bin/bash
# round 4, GPU session 25: the scheduling constants of the own-leaf kernels once more on the cheaper box step and refill (refill at 44):
# how long the box stream keeps going (1/8 + 1/6, 1/4 + 1/3 of its starters instead of 1/6 + 1/4), 6 leaves per vote, leaf stream to 1/2
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s25; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_own_leaves.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2 3; do
  TAG="cfg1 base (refill 44, keep 6/4)" b --overlap 0
  for v in nk8 nk4 ls6 lk2; do TAG="cfg1 $v" PTMI_LIB=$ab/libptmi_$v.so b --overlap 0; done
done 2>&1 | tee $out/ab_cfg1_one_stream.txt
for round in 1 2; do
  TAG="cfg1 base" b
  for v in nk8 nk4 ls6 lk2; do TAG="cfg1 $v" PTMI_LIB=$ab/libptmi_$v.so b; done
  TAG="cfg2 base" b --config 2 --steps 2
  for v in nk8 nk4 ls6 lk2; do TAG="cfg2 $v" PTMI_LIB=$ab/libptmi_$v.so b --config 2 --steps 2; done
done 2>&1 | tee $out/ab_two_streams.txt
