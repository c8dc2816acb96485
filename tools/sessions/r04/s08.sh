#!/bin/bash
# round 4, GPU session 8: the whole GPU suite on the final sources, over the library's own leaves (the default) and over the reference's
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s08; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_own_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_own_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_own_leaves.log | head -100; exit 1; }
PTMI_TEST_LEAVES=1 timeout -k 10 560 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_reference_leaves.log 2>&1; rc=$?; tail -3 $out/pytest_gpu_reference_leaves.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu_reference_leaves.log | head -100; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()"
