#!/bin/bash
scene=$1; root=$PWD; out=$root/gpurun_out/pmc_$scene; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $root
python3 bench.py --no-cpu-baseline --timing 3 --scene $scene ${2:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$scene', d['value'], d['kernel_ms_rank0'], d['segments'], d['shadow_rays'])"
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/g1 -- python3 bench.py --no-cpu-baseline --scene $scene ${2:-} > /dev/null 2> $out/g1.err || echo failed
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --scene $scene ${2:-} > /dev/null 2>> $out/g1.err
python3 tools/pmc_summary.py --json $(find $out/g1 -name "*counter_collection.csv") > $out/util.json
grep -E "k_trace|k_shade" $(find $out/kt -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-160
python3 - <<PY
import json
d=json.load(open('$out/util.json'))
for k in sorted(d.get('SQ_ACTIVE_INST_VALU',{})):
    if 'trace' in k or 'shade' in k:
        a=d['SQ_ACTIVE_INST_VALU'][k]['avg_per_launch']; t=d['SQ_THREAD_CYCLES_VALU'][k]['avg_per_launch']; v=d['SQ_INSTS_VALU'][k]['avg_per_launch']
        print(f"{k:22s} util {t/(a*64):.3f} VALU/launch {v:.3e} launches {d['SQ_INSTS_VALU'][k]['launches']}")
PY
rm -rf $out/g1 $out/kt
