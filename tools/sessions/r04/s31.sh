#!/bin/bash
# round 4, GPU session 31: every memory variant of the own-leaf kernels forced in turn (where it fits; elsewhere the library's own choice runs),
# BASELINE configs at full sample counts in both leaf modes: frames equal bit for bit
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s31; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
for v in 4 5 15 6 7 17 8 9 20 21; do
  echo "== PTMI_OWN_EXTEND = PTMI_OWN_SHADOW = $v"
  PTMI_OWN_EXTEND=$v PTMI_OWN_SHADOW=$v timeout -k 10 300 python tools/leaf_modes_equal_gpu.py 1 || exit 1
done 2>&1 | tee $out/variants_equal.log | grep "==\|total"
