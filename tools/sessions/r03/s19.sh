#!/bin/bash
# round 3, GPU session 19: where does the prefetched raygen run? kernel trace of three pipelined steps of config 1
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s19; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --config 1 --steps 3 --pipeline 2 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 150 > $out/timeline_cfg1_pipeline.txt
grep -n "k_raygen\|k_accumulate\|extend_lds       #[0-9]*[08] \|k_shade          #[0-9]*7 " $out/timeline_cfg1_pipeline.txt | head -40
rm -rf $out/kt
