#!/bin/bash
# GPU session 51: the driver's own command line
set -o pipefail
out=gpurun_out/s51; mkdir -p $out
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; rc=$?; [ $rc = 0 ] || { tail -5 $out/bench_driver.err; exit $rc; }
python3 -c "
import json; d=json.load(open('$out/bench_driver.json')); r=d['roofline']
print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], r['kernel'], r['frac'], r['traffic'], r['valu_issue']['frac'] if r.get('valu_issue') else None, d['cpu_baseline']['value'], d['kernel_ms_sum_over_gpu_ms'])"
