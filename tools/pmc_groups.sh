#!/bin/bash
# tools/pmc_groups.sh <tag> "<group 1 counters>" "<group 2 counters>" ... -- [bench flags...]: one rocprofv3 --pmc pass of bench.py per
# counter group (counters in their own runs, no trace domains), per-kernel means into gpurun_out/pmc_<tag>/summary.json
tag=$1; shift
groups=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do groups+=("$1"); shift; done; [ "$1" = "--" ] && shift
root=$PWD; out=$root/gpurun_out/pmc_$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $root
i=0
for grp in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --no-cpu-baseline "$@" > /dev/null 2> $out/g$i.err || echo "group $i failed: $grp"
done
python3 tools/pmc_summary.py --json ${PT_COMMIT:+--commit $PT_COMMIT} $(find $out -name "*counter_collection.csv") > $out/summary.json
rm -rf $out/g[0-9]
