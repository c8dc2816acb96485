#!/bin/bash
# tools/pmc_util.sh: VALU lane utilisation per kernel = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64), plus instruction counts
root=$PWD; out=$root/gpurun_out/pmc_util; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $root
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/g1 -- python3 bench.py --no-cpu-baseline > /dev/null 2> $out/g1.err || echo failed
python3 tools/pmc_summary.py --json $(find $out -name "*counter_collection.csv") > $out/util.json
rm -rf $out/g1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_util/util.json'))
for k in sorted(d.get('SQ_ACTIVE_INST_VALU',{})):
    a=d['SQ_ACTIVE_INST_VALU'][k]['avg_per_launch']; t=d['SQ_THREAD_CYCLES_VALU'][k]['avg_per_launch']
    v=d['SQ_INSTS_VALU'][k]['avg_per_launch']; s=d['SQ_INSTS_SALU'][k]['avg_per_launch']
    print(f"{k:22s} lane utilisation {t/(a*64) if a else 0:.3f}  VALU/launch {v:.3e}  SALU/launch {s:.3e}  active cycles per VALU {a/v if v else 0:.2f}")
PY
