#!/bin/bash
# GPU session 1 of round 2: tests, upload times, baseline of every config, A/B of the leaf cull
set -o pipefail
out=gpurun_out/s1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; [ $rc = 0 ] || exit $rc
python tools/time_upload.py > $out/upload.log 2>&1 && cat $out/upload.log &&
tools/ubench/pk.bin > $out/pk.log 2>&1 && cat $out/pk.log &&
for c in 1 2 3 4; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err || exit 1; python -c "
import json; d=json.load(open('$out/bench_cfg$c.json')); print('cfg$c', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['kernel_ms_sum_over_gpu_ms'], d['upload_ms_rank0'], d['config']['traversal'])"; done &&
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_nolc.so wgpu-path-tracing_amd/lib/libptmi.so 2 --config 1 > $out/ab_lc_cfg1.log 2>&1 && cat $out/ab_lc_cfg1.log &&
bash tools/ab.sh wgpu-path-tracing_amd/lib/ab/libptmi_nolc.so wgpu-path-tracing_amd/lib/libptmi.so 1 --config 2 --steps 4 > $out/ab_lc_cfg2.log 2>&1 && cat $out/ab_lc_cfg2.log
