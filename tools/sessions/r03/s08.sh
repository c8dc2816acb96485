#!/bin/bash
# round 3, GPU session 8: the timeline of one overlapped step of config 1 (what runs beside what, where streams wait)
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s08; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --config 1 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 60 > $out/timeline_cfg1.txt
cat $out/timeline_cfg1.txt
rm -rf $out/kt
