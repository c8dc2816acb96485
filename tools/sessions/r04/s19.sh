#!/bin/bash
# round 4, GPU session 19: the frozen sources — large random scenes against the oracle, more edge cases, 30 soak rounds (same dispatches,
# same bits and counters every time)
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s19; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 600 python tools/fuzz_big_gpu.py > $out/fuzz_big.log 2>&1; rc=$?; tail -3 $out/fuzz_big.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/edge_more_gpu.py > $out/edge_more.log 2>&1; rc=$?; tail -3 $out/edge_more.log; [ $rc = 0 ] || exit $rc
timeout -k 10 500 python tools/soak_gpu.py 30 > $out/soak30.log 2>&1; rc=$?; tail -3 $out/soak30.log; exit $rc
