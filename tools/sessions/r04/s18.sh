#!/bin/bash
# round 4, GPU session 18: the frozen sources — the whole GPU suite in both leaf modes, smoke, then the heavier seeded fuzzers
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s18; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_own_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_own_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_own_leaves.log | head -100; exit 1; }
PTMI_TEST_LEAVES=1 timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_reference_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_reference_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_reference_leaves.log | head -100; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" && echo smoke ok
timeout -k 10 400 python tools/fuzz_gpu.py 20 256 192 4 > $out/fuzz_gpu.log 2>&1; rc=$?; tail -3 $out/fuzz_gpu.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/fuzz_scale_gpu.py > $out/fuzz_scale_gpu.log 2>&1; rc=$?; tail -3 $out/fuzz_scale_gpu.log; [ $rc -ne 0 ] && exit 1
echo done
