#!/bin/bash
# tools/profile_round.sh <tag>: the four runs behind profiles/<tag>_bench_n1*. Run on the GPU box from the repo root
# (gpurun -- 'tools/profile_round.sh r01'); results land in gpurun_out/prof_<tag>/.
set -e
tag=${1:-r01}; root=$PWD; out=$root/gpurun_out/prof_$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py > $out/bench_profiled.json 2>> $out/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu-baseline > /dev/null 2>> $out/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu-baseline > /dev/null 2>> $out/bench.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 tools/pmc_summary.py --json $(find $out/pmc_fetch $out/pmc_write -name "*counter_collection.csv") > $out/pmc.json
rm -rf $out/kt $out/pmc_fetch $out/pmc_write
cat $out/bench.json; head -12 $out/kernel_stats.csv
