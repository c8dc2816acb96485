#!/bin/bash
# round 3, GPU session 12: the committed evidence of the round (profiles/r03_*; $1 = the commit of the kernel sources): upload times, the multi-device handle's gather, then
# per config the counter passes, the bench lines, the kernel trace (tools/profile_round.sh), lane statistics from the diagnostic build
set -o pipefail
commit=${1:?usage: tools/sessions/r03/s12.sh <commit of the kernel sources in this snapshot>}
mkdir -p gpurun_out/s12
python tools/time_upload.py > gpurun_out/s12/upload.log 2>&1; cat gpurun_out/s12/upload.log
timeout -k 10 300 python tools/multi_gather_time.py gpurun_out/s12/multi_gather.json > gpurun_out/s12/multi_gather.out 2> gpurun_out/s12/multi_gather.err || tail -5 gpurun_out/s12/multi_gather.err; cat gpurun_out/s12/multi_gather.json
for c in 1 3 2 4; do timeout -k 10 700 bash tools/profile_round.sh r03 $c $commit > gpurun_out/prof_r03_cfg$c.log 2>&1 || { tail -5 gpurun_out/prof_r03_cfg$c.log; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r03/r03_cfg${c}_bench.json')); print('cfg$c', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'])"; done
if [ -f wgpu-path-tracing_amd/lib/ab/libptmi_util.so ]; then for c in 1 2 3; do PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_util.so timeout -k 10 300 python3 tools/lane_stats.py $c > gpurun_out/prof_r03/r03_cfg${c}_lane_stats.json 2> gpurun_out/prof_r03/lanes$c.err || exit 1; done; fi
python3 bench.py --config 0 > gpurun_out/prof_r03/r03_cfg0_bench.json 2> gpurun_out/prof_r03/r03_cfg0.err; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r03/r03_cfg0_bench.json')); print('cfg0', d['value'], d['kernel_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'])"
