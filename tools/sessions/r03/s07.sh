#!/bin/bash
# round 3, GPU session 7: why is a late bounce's segment 2 - 5 x dearer in `shade` than a bounce-0 one, and the state following the
# queue no help? Per-bounce counters (one stream, so that a launch's counters are its own): instructions, lane cycles, busy and
# wait cycles, memory requests and bytes — for the state in place and following the queue
set -o pipefail
root=$PWD; cd /tmp; export TMPDIR=/tmp; cd $root
g1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
g2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
g3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
g4="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
g5="FETCH_SIZE"
g6="WRITE_SIZE"
g7="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
g8="SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
for st in 1 2; do
  out=$root/gpurun_out/r03_s07/state$st; mkdir -p $out
  i=0
  for grp in "$g1" "$g2" "$g3" "$g4" "$g5" "$g6" "$g7" "$g8"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --no-cpu-baseline --config 1 --overlap 0 --state $st > /dev/null 2> $out/g$i.err || echo "group $i failed: $grp"
  done
  python3 tools/pmc_per_bounce.py $(find $out -name "*counter_collection.csv") > $root/gpurun_out/r03_s07/per_bounce_state$st.json
  rm -rf $out/g[0-9]
done
python3 - <<'PY'
import json
for st in (1, 2):
    d = json.load(open(f'gpurun_out/r03_s07/per_bounce_state{st}.json'))
    for c in d:
        print(st, c, 'shade', d[c].get('shade'))
PY
