#!/bin/bash
# round 3, GPU session 3: is the LDS array the traversal kernels' second limit? LDS-array cycles and bank conflicts beside the VALU
# counters, for the default build (extend from the node cache), both kernels from the full LDS image, and the work-list variant
set -o pipefail
g1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
g2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
g3="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
g4="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
g5="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SALU"
tools/pmc_groups.sh r03_lds_auto "$g1" "$g2" "$g3" "$g4" "$g5" -- --config 1
tools/pmc_groups.sh r03_lds_off "$g1" "$g2" "$g3" "$g4" "$g5" -- --config 1 --traversal lds --worklist 1
tools/pmc_groups.sh r03_lds_wl "$g1" "$g2" "$g3" "$g4" "$g5" -- --config 1 --traversal lds --worklist 2
python3 - <<'PY'
import json
for tag in ("auto", "off", "wl"):
    d = json.load(open(f"gpurun_out/pmc_r03_lds_{tag}/summary.json"))
    for k in ("k_trace_lds/extend", "k_trace_lds/shadow"):
        g = lambda c: d.get(c, {}).get(k, {}).get("avg_per_launch")
        print(tag, k, {c: g(c) for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES")})
PY
