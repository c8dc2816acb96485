#!/bin/bash
# round 4, GPU session 13: the two streams' persistent grids sized to shares of the CUs (both resident at once, no waiting for slots)
# against full-size grids that take turns; -DPT_GRID_SPLIT_AB build, PTMI_EXTEND_CU_PCT / PTMI_SHADOW_CU_PCT
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s13; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
export PTMI_LIB=$root/wgpu-path-tracing_amd/lib/ab/libptmi_split.so
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
for round in 1 2; do
  TAG="cfg1 100 / 100" b
  for es in "100 50" "100 62" "100 75" "75 50" "75 25" "62 38" "50 50" "100 25"; do set -- $es
    TAG="cfg1 extend $1 / shadow $2" PTMI_EXTEND_CU_PCT=$1 PTMI_SHADOW_CU_PCT=$2 b
  done
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 100 / 100" b --config 2 --steps 1
  for es in "100 50" "75 50" "62 38"; do set -- $es
    TAG="cfg2 extend $1 / shadow $2" PTMI_EXTEND_CU_PCT=$1 PTMI_SHADOW_CU_PCT=$2 b --config 2 --steps 1
  done
  TAG="cfg3 100 / 100" b --config 3 --steps 1
  for es in "100 50" "75 50" "62 38"; do set -- $es
    TAG="cfg3 extend $1 / shadow $2" PTMI_EXTEND_CU_PCT=$1 PTMI_SHADOW_CU_PCT=$2 b --config 3 --steps 1
  done
done 2>&1 | tee $out/ab_cfg23.txt
