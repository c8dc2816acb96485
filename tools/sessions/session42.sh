#!/bin/bash
# GPU session 42: what the driver runs at round end — smoke(), then bench.py with no flags and with --steps 5 --warmup 2
set -o pipefail
out=gpurun_out/s42; mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; rc=$?; tail -2 $out/smoke.log; [ $rc = 0 ] || exit $rc
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; rc=$?; [ $rc = 0 ] || { tail -5 $out/bench_default.err; exit $rc; }
python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu_issue']['frac'] if d['roofline'].get('valu_issue') else None, d['cpu_baseline']['value'])"
timeout -k 10 400 python bench.py --gpus 1 --steps 5 --warmup 2 > $out/bench_5.json 2> $out/bench_5.err; rc=$?; [ $rc = 0 ] || { tail -5 $out/bench_5.err; exit $rc; }
python -c "
import json; d=json.load(open('$out/bench_5.json')); print('steps5', d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu_issue'))"
