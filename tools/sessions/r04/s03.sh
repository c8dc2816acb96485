#!/bin/bash
# round 4, GPU session 3: config 1 A/Bs on one box — the refill threshold of the own-leaf kernels, the any-hit kernel from two
# workgroups per CU, shade_sort; shade_sort on configs 2 and 3
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s03; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shade_sort or render_parity or overlapped" > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], d['kernel_ms_rank0'])"; }
for round in 1 2; do
  TAG="base         " b
  TAG="shade_sort 2 " b --shade-sort 2
  TAG="shadow 2wg q " PTMI_OWN_SHADOW=17 b
  TAG="shadow 2wg q + sort" PTMI_OWN_SHADOW=17 b --shade-sort 2
  TAG="refill 44    " PTMI_LIB=$root/wgpu-path-tracing_amd/lib/ab/libptmi_refill44.so b
  TAG="refill 52    " PTMI_LIB=$root/wgpu-path-tracing_amd/lib/ab/libptmi_refill52.so b
  TAG="leaves 1     " b --leaves 1
  TAG="leaves 1 sort" b --leaves 1 --shade-sort 2
done 2>&1 | tee $out/ab_cfg1.txt
for cfg in 2 3; do
  for round in 1 2; do
    TAG="cfg$cfg base  " b --config $cfg --steps 2
    TAG="cfg$cfg sort 2" b --config $cfg --steps 2 --shade-sort 2
  done
done 2>&1 | tee $out/ab_cfg23.txt
TAG="one stream base" b --overlap 0
TAG="one stream sort" b --overlap 0 --shade-sort 2
