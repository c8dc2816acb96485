#!/bin/bash
# GPU session 31: lane statistics of the traversal kernels (diagnostic build with -DPT_UTIL_STATS), configs 1, 2, 3
set -o pipefail
out=gpurun_out/s31; mkdir -p $out
export PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_util.so
for c in 1 2 3; do timeout -k 10 300 python tools/lane_stats.py $c > $out/lanes_cfg$c.json 2> $out/lanes_cfg$c.err || { tail -5 $out/lanes_cfg$c.err; exit 1; }; cat $out/lanes_cfg$c.json; done
