#!/bin/bash
# round 3, GPU session 15: a cycle account of the quantised traversal kernels on the 1 M-triangle scene, one stream (a launch's counters are
# its own): what the waves issue (VALU / scalar / LDS / VMEM / misc), what they wait for, instruction fetch — beside the LDS kernels of config 1
set -o pipefail
g1="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
g2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU"
g3="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
g4="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU"
g5="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH"
g6="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD"
tools/pmc_groups.sh r03_acct_cfg3 "$g1" "$g2" "$g3" "$g4" "$g5" "$g6" -- --config 3 --overlap 0
tools/pmc_groups.sh r03_acct_cfg1 "$g1" "$g2" "$g3" "$g4" "$g5" "$g6" -- --config 1 --overlap 0
python3 - <<'PY'
import json
for tag in ("cfg3", "cfg1"):
    d = json.load(open(f"gpurun_out/pmc_r03_acct_{tag}/summary.json"))
    ks = sorted({k for c in d.values() if isinstance(c, dict) for k in c if 'trace' in k or 'shade' in k})
    for k in ks:
        g = lambda c: d.get(c, {}).get(k, {}).get("avg_per_launch")
        print(tag, k, {c: g(c) for c in sorted(d) if not c.startswith('_') and g(c) is not None})
PY
