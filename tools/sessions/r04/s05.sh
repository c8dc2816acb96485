#!/bin/bash
# round 4, GPU session 5: interleaved batches — the full GPU suite, then same-box A/Bs (PTMI_INTERLEAVE=0 is the sequential loop) on
# configs 1 (8 steps), 2 (512 spp = 8 batches) and 3 (4 steps); VALU counters of shade with and without shade_sort
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s05; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -4 $out/pytest_gpu.log
[ $rc -ne 0 ] && { grep -B5 -A40 "Error\|FAILED" $out/pytest_gpu.log | head -120; exit 1; }
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', d['value'], d['ms_per_step'], d['config']['batches_interleaved'], d['config']['frames_per_batch'], d['kernel_ms_rank0'])"; }
for round in 1 2; do
  TAG="cfg1 x8 interleaved " b --steps 8
  TAG="cfg1 x8 sequential  " PTMI_INTERLEAVE=0 b --steps 8
  TAG="cfg1 x8 per step    " b --steps 8 --dispatch-per-step
  TAG="cfg1 x8 inter fpb 32" b --steps 8 --frames-per-batch 32
  TAG="cfg1 x1             " b
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 interleaved " b --config 2
  TAG="cfg2 sequential  " PTMI_INTERLEAVE=0 b --config 2
  TAG="cfg3 x4 interleaved" b --config 3 --steps 4
  TAG="cfg3 x4 sequential " PTMI_INTERLEAVE=0 b --config 3 --steps 4
done 2>&1 | tee $out/ab_cfg23.txt
for ss in 1 2; do
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc$ss -- python3 bench.py --no-cpu-baseline --no-leaves-compare --overlap 0 --steps 1 --shade-sort $ss > /dev/null 2> $out/pmc$ss.err || echo "pmc $ss failed"
  python3 tools/pmc_summary.py --json $(find $out/pmc$ss -name "*counter_collection.csv") > $out/pmc_shade_sort$ss.json; rm -rf $out/pmc$ss
  python3 -c "
import json
d = json.load(open('$out/pmc_shade_sort$ss.json'))
k = 'k_shade'
g = lambda c: d.get(c, {}).get(k, {}).get('avg_per_launch', 0)
print('shade_sort $ss', k, 'launches', d['SQ_INSTS_VALU'][k]['launches'], 'VALU insts/launch %.4g' % g('SQ_INSTS_VALU'), 'SALU %.4g' % g('SQ_INSTS_SALU'), 'lane util %.3f' % (g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU') / 64), 'busy ms/launch at 2.4 GHz %.3f' % (4 * g('SQ_ACTIVE_INST_VALU') / 1024 / 2.4e6))"
done 2>&1 | tee $out/shade_sort_counters.txt
