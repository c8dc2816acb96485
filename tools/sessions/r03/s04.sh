#!/bin/bash
# round 3, GPU session 4: work list, second version (box steps, listing and testing in one iteration): parity, A/B, lane statistics
set -o pipefail
out=gpurun_out/r03_s04; mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_worklist.py -m gpu -x -q > $out/pytest_wl.log 2>&1; rc=$?; tail -5 $out/pytest_wl.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('worklist_used'))"; }
for i in 1 2; do
  run lds_off_one_$i --config 1 --traversal lds --worklist 1 --overlap 0 && run lds_wl_one_$i --config 1 --traversal lds --worklist 2 --overlap 0 || exit 1
done
run lds_off --config 1 --traversal lds --worklist 1 && run lds_wl --config 1 --traversal lds --worklist 2 && run auto --config 1 || exit 1
ab=$PWD/wgpu-path-tracing_amd/lib/ab
PTMI_LIB=$ab/libptmi_util.so PTMI_OPTS='{"traversal": 2, "worklist": 2}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_lds_wl.json 2> $out/lane_wl.err || tail -3 $out/lane_wl.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s04/lane_lds_wl.json'))
for k in ('extend','shadow'): print(k, {x: d[k][x] for x in ('votes_per_ray_x64','lanes_holding_a_ray_at_vote','refills_per_64_rays','box_step_lane_util','leaf_open_lane_util','triangle_lane_util','wave_steps_per_64_rays')})
PY
