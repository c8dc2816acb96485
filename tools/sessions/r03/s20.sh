#!/bin/bash
# round 3, GPU session 20: the prefetched raygen on a LOWEST-priority stream (its own hardware queue): where it runs now, parity, A/B
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s20; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --config 1 --steps 3 --pipeline 2 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 150 > $out/timeline_cfg1_pipeline.txt
grep -n "k_raygen\|k_accumulate" $out/timeline_cfg1_pipeline.txt | head; awk '{print $3,$4}' $out/timeline_cfg1_pipeline.txt | sort | uniq -c
rm -rf $out/kt
timeout -k 10 600 python -m pytest tests/test_gpu_state.py -m gpu -x -q > $out/pytest_state.log 2>&1; rc=$?; tail -3 $out/pytest_state.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['config'].get('pipeline_used'))"; }
def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so; nh=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_nohold.so
for i in 1 2 3; do
  run c1_off_$i $def --config 1 --steps 8 --pipeline 1 && run c1_on_$i $def --config 1 --steps 8 --pipeline 2 && run c1_nohold_$i $nh --config 1 --steps 8 --pipeline 2 || exit 1
done
for i in 1 2; do
  run c3_off_$i $def --config 3 --steps 6 --pipeline 1 && run c3_on_$i $def --config 3 --steps 6 --pipeline 2 && run c3_nohold_$i $nh --config 3 --steps 6 --pipeline 2 || exit 1
  run c4_off_$i $def --config 4 --pipeline 1 && run c4_on_$i $def --config 4 --pipeline 2 || exit 1
  run c2_off_$i $def --config 2 --pipeline 1 && run c2_on_$i $def --config 2 --pipeline 2 || exit 1
done
