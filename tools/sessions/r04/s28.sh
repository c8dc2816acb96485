#!/bin/bash
# round 4, GPU session 28: the final sources — the bench command of config 1 twelve times in a row on one box (how much one box moves
# between runs), then 20 soak rounds
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s28; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; lc=d['leaves_compare']; print('run $i', d['value'], 'reference leaves', lc['value'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"
done 2>&1 | tee $out/cfg1_twelve_runs.txt
timeout -k 10 500 python tools/soak_gpu.py 20 > $out/soak20.log 2>&1; rc=$?; tail -2 $out/soak20.log; exit $rc
