#!/bin/bash
# round 3, GPU session 25: the shadow stream's hardware queue. Two contexts alive at once (a plain one and a multi-device handle with RCCL
# loaded) showed the second one 5.7 % slower — the one-stream figure: its shadow stream shares a hardware queue with its main stream.
# Streams of another PRIORITY have hardware queues of their own: the shadow stream at high / low priority against the default (normal)
set -o pipefail
out=gpurun_out/r03_s25; mkdir -p $out
ab=$PWD/wgpu-path-tracing_amd/lib/ab; def=$PWD/wgpu-path-tracing_amd/lib/libptmi.so
for v in def sidehigh sidelow; do
  lib=$def; [ $v != def ] && lib=$ab/libptmi_$v.so
  PTMI_LIB=$lib timeout -k 10 300 python tools/multi_vs_plain.py $out/multi_vs_plain_$v.json > /dev/null 2> $out/mvp_$v.err || { tail -3 $out/mvp_$v.err; exit 1; }
  python -c "import json; d=json.load(open('$out/multi_vs_plain_$v.json')); print('$v', d)"
done
run() { tag=$1; lib=$2; shift 2; PTMI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_rank0'])"; }
for i in 1 2 3; do
  run c1_def_$i $def --config 1 --steps 4 && run c1_high_$i $ab/libptmi_sidehigh.so --config 1 --steps 4 && run c1_low_$i $ab/libptmi_sidelow.so --config 1 --steps 4 || exit 1
done
for i in 1 2; do
  run c3_def_$i $def --config 3 --steps 2 && run c3_high_$i $ab/libptmi_sidehigh.so --config 3 --steps 2 && run c3_low_$i $ab/libptmi_sidelow.so --config 3 --steps 2 || exit 1
done
run c2_def $def --config 2 --steps 2 && run c2_high $ab/libptmi_sidehigh.so --config 2 --steps 2
