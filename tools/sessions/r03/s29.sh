#!/bin/bash
# round 3, GPU session 29: per-bounce counters (one stream) for configs 3 and 2 — the same account session 7 took for config 1 — and a
# 40-round soak of the final build
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
g1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
g2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
g3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
g4="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
g5="FETCH_SIZE"
g6="WRITE_SIZE"
for cfg in 3 2; do
  out=$root/gpurun_out/r03_s29/cfg$cfg; mkdir -p $out
  i=0
  for grp in "$g1" "$g2" "$g3" "$g4" "$g5" "$g6"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --no-cpu-baseline --config $cfg --overlap 0 --steps 1 > /dev/null 2> $out/g$i.err || echo "group $i failed: $grp"
  done
  python3 tools/pmc_per_bounce.py $(find $out -name "*counter_collection.csv") > $root/gpurun_out/r03_s29/per_bounce_cfg${cfg}_one_stream.json
  rm -rf $out/g[0-9]
done
timeout -k 10 600 python tools/soak_gpu.py 40 > $root/gpurun_out/r03_s29/soak40.log 2>&1; tail -2 $root/gpurun_out/r03_s29/soak40.log
python3 - <<'PY'
import json
for cfg in (3, 2):
    d = json.load(open(f'gpurun_out/r03_s29/per_bounce_cfg{cfg}_one_stream.json'))
    for k in ('extend', 'shade', 'shadow'):
        v = [x * 4 / (1024 * 2.4e9) * 1e3 for x in d['SQ_ACTIVE_INST_VALU'][k]]
        t = [x / 8 / 2.4e9 * 1e3 for x in d['GRBM_GUI_ACTIVE'][k]]
        lu = [a / (b * 64) if b else 0 for a, b in zip(d['SQ_THREAD_CYCLES_VALU'][k], d['SQ_ACTIVE_INST_VALU'][k])]
        print(cfg, k, 'VALU ms', [round(x, 2) for x in v], 'time ms', [round(x, 2) for x in t], 'lanes', [round(x, 2) for x in lu])
PY
