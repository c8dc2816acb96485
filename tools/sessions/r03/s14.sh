#!/bin/bash
# round 3, GPU session 14: is PC sampling available here? (where do the quantised kernel's waves wait) — a probe, bounded
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
out=$root/gpurun_out/r03_s14; mkdir -p $out
rocprofv3 --help 2>&1 | grep -i -A3 "pc-sampling" | head -40 > $out/help.txt; cat $out/help.txt
rocprofv3 -L 2>/dev/null | grep -i -E "IFETCH|INST_LEVEL|WAIT_IFETCH|SQ_WAIT|SQ_INST_CYCLES|SQ_ACTIVE_INST|SQ_LEVEL|VALU_DEP|STALL" | head -60 > $out/counters.txt; cat $out/counters.txt
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 200 --kernel-trace --output-format csv -d $out/pcs -- python3 bench.py --no-cpu-baseline --config 3 --frames-per-step 8 --steps 1 --warmup 1 > $out/bench.json 2> $out/pcs.err; echo "rc $?"
tail -5 $out/pcs.err
find $out/pcs -type f | head; du -sh $out/pcs 2>/dev/null
