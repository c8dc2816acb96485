#!/bin/bash
# tools/pmc_lds.sh: LDS / VALU balance counters of the traversal kernels (one rocprofv3 --pmc pass per group)
root=$PWD; out=$root/gpurun_out/pmc_lds; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $root
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --no-cpu-baseline > /dev/null 2> $out/g$i.err || echo "group $i failed: $grp"
done
python3 tools/pmc_summary.py $(find $out -name "*counter_collection.csv") > $out/summary.txt 2>&1
rm -rf $out/g[0-9]
grep -A 16 "k_trace_lds" $out/summary.txt | head -60
