#!/bin/bash
# GPU session 32: flat triangle stream — whole GPU suite on it, lane statistics, then A/B against the per-leaf loop
set -o pipefail
out=gpurun_out/s32; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_util.so timeout -k 10 300 python tools/lane_stats.py 1 > $out/lanes_cfg1.json 2> $out/lanes.err || { tail -5 $out/lanes.err; exit 1; }
python -c "
import json; d=json.load(open('$out/lanes_cfg1.json'))
for k in ('extend','shadow'): print(k, d[k]['triangle_lane_util'], d[k]['box_step_lane_util'], d[k]['wave_steps_per_64_rays'])"
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do
  PTMI_LIB=$ab/libptmi_noflat.so run cfg1_perleaf_$i --config 1 && run cfg1_flat12_$i --config 1 && PTMI_LIB=$ab/libptmi_tri8.so run cfg1_flat8_$i --config 1 && PTMI_LIB=$ab/libptmi_tri16.so run cfg1_flat16_$i --config 1 || exit 1
done
PTMI_LIB=$ab/libptmi_noflat.so run cfg1_perleaf_one --config 1 --overlap 0 && run cfg1_flat12_one --config 1 --overlap 0
for i in 1 2; do PTMI_LIB=$ab/libptmi_noflat.so run cfg2_perleaf_$i --config 2 && run cfg2_flat12_$i --config 2 || exit 1; done
for i in 1 2; do PTMI_LIB=$ab/libptmi_noflat.so run cfg3_perleaf_$i --config 3 && run cfg3_flat12_$i --config 3 || exit 1; done
