#!/bin/bash
# round 3, GPU session 28: the shadow stream at HIGH priority (the new default) against normal priority (-DPT_SIDE_NORMAL_PRIORITY):
# a second context of a process (tools/two_contexts.py a / b, un-profiled), the bench lines interleaved, then the whole GPU suite
set -o pipefail
out=gpurun_out/r03_s28; mkdir -p $out
def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so; nrm=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_sidenormal.so
for mode in a b a b; do
  echo "high   $(PTMI_LIB=$def timeout -k 10 200 python tools/two_contexts.py $mode 2>/dev/null | tail -1)" | tee -a $out/two_contexts.txt
  echo "normal $(PTMI_LIB=$nrm timeout -k 10 200 python tools/two_contexts.py $mode 2>/dev/null | tail -1)" | tee -a $out/two_contexts.txt
done
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  run c1_high_$i $def --config 1 --steps 4 && run c1_normal_$i $nrm --config 1 --steps 4 || exit 1
done
run c3_high $def --config 3 --steps 2 && run c3_normal $nrm --config 3 --steps 2 && run c2_high $def --config 2 --steps 2 && run c2_normal $nrm --config 2 --steps 2 && run c4_high $def --config 4 --steps 1 && run c4_normal $nrm --config 4 --steps 1 || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -3 $out/pytest_gpu.log; exit $rc
