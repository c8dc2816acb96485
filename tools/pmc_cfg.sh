#!/bin/bash
# tools/pmc_cfg.sh <config> <tag> [bench flags...]: counter passes of bench.py --config N beyond FETCH/WRITE_SIZE — VALU lane
# utilisation, instruction counts, cache hit rates — one rocprofv3 --pmc run per group (gpurun_out/pmc_<tag>/summary.json)
cfg=$1; tag=$2; shift 2
root=$PWD; out=$root/gpurun_out/pmc_$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $root
i=0
for grp in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --config $cfg --no-cpu-baseline --no-leaves-compare "$@" > /dev/null 2> $out/g$i.err || echo "group $i failed: $grp"
done
python3 tools/pmc_summary.py --json ${PT_COMMIT:+--commit $PT_COMMIT} $(find $out -name "*counter_collection.csv") > $out/summary.json
rm -rf $out/g[0-9]
python3 - <<PY
import json
d=json.load(open('$out/summary.json'))
ks=sorted({k for c in d.values() if isinstance(c,dict) for k in c})
for k in ks:
    if not ('trace' in k or 'own' in k or 'shade' in k): continue
    g=lambda c: d.get(c,{}).get(k,{}).get('avg_per_launch')
    a,t,v,s=g('SQ_ACTIVE_INST_VALU'),g('SQ_THREAD_CYCLES_VALU'),g('SQ_INSTS_VALU'),g('SQ_INSTS_SALU')
    line=f"{k:24s}"
    if a and t: line+=f" lane util {t/(a*64):.3f} VALU {v:.3e} SALU {s:.3e}"
    for c in ('SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_INST_ANY','SQ_INST_CYCLES_VMEM','TCP_TOTAL_CACHE_ACCESSES_sum','TCP_TCC_READ_REQ_sum','TCP_PENDING_STALL_CYCLES_sum','TCC_HIT_sum','TCC_MISS_sum','TCC_EA0_RDREQ_sum','TCC_EA0_RDREQ_32B_sum','SQ_INSTS_VMEM_RD','SQ_INSTS_LDS'):
        x=g(c)
        if x is not None: line+=f" {c.replace('_sum','')}={x:.3e}"
    print(line)
PY
