#!/bin/bash
# GPU session 15: whole suite on the final build, then the committed evidence again (profiles/r02_*: counters, bench lines,
# kernel trace per config; lane statistics from the diagnostic build when lib/ab/libptmi_util.so is there)
set -o pipefail
mkdir -p gpurun_out/s15
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s15/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/s15/pytest.log; [ $rc = 0 ] || exit $rc
python tools/time_upload.py > gpurun_out/s15/upload.log 2>&1; cat gpurun_out/s15/upload.log
for c in 1 3 2 4; do timeout -k 10 600 bash tools/profile_round.sh r02 $c 6536d71 > gpurun_out/prof_r02_cfg$c.log 2>&1 || { tail -5 gpurun_out/prof_r02_cfg$c.log; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r02/r02_cfg${c}_bench.json')); print('cfg$c', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['roofline']['frac'])"; done
if [ -f wgpu-path-tracing_amd/lib/ab/libptmi_util.so ]; then for c in 1 2 3; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_util.so timeout -k 10 300 python3 tools/lane_stats.py $c > gpurun_out/prof_r02/r02_cfg${c}_lane_stats.json 2> gpurun_out/prof_r02/lanes$c.err || exit 1; done; fi
python3 bench.py --config 0 > gpurun_out/prof_r02/r02_cfg0_bench.json 2> gpurun_out/prof_r02/r02_cfg0.err; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r02/r02_cfg0_bench.json')); print('cfg0', d['value'], d['kernel_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'])"
