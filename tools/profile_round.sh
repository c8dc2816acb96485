#!/bin/bash
# tools/profile_round.sh <tag> <config> [code-commit] [extra bench flags...]: the runs behind profiles/<tag>_cfg<config>_*.
# Run on the GPU box from the repo root (gpurun -- 'tools/profile_round.sh r02 1 abc1234'); results land in
# gpurun_out/prof_<tag>/ under the names they are committed with:
#   <tag>_cfgN_bench.json         python3 bench.py --config N                                  (un-profiled line, with cpu_baseline)
#   <tag>_cfgN_kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python3 bench.py --config N
#   <tag>_cfgN_bench_profiled.json  the line that same profiled run printed (its HIP-event launch means must agree with the CSV)
#   <tag>_cfgN_pmc.json           separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, mean KB per launch per kernel
#   <tag>_cfgN_counters.json      tools/pmc_cfg.sh: SQ / TCP / TCC counters per kernel, one --pmc run per group
set -e
tag=${1:-r02}; cfg=${2:-1}; commit=${3:-unknown}; shift 3 || true
root=$PWD; out=$root/gpurun_out/prof_$tag; pre=$out/${tag}_cfg${cfg}
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
cd $root
# counters first, copied into profiles/ of this snapshot, so that the bench lines made below replay the counters of THIS build
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf$cfg -- python3 bench.py --config $cfg --no-cpu-baseline --no-leaves-compare "$@" > /dev/null 2> ${pre}.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw$cfg -- python3 bench.py --config $cfg --no-cpu-baseline --no-leaves-compare "$@" > /dev/null 2>> ${pre}.err
python3 tools/pmc_summary.py --json --commit $commit $(find $out/pf$cfg $out/pw$cfg -name "*counter_collection.csv") > ${pre}_pmc.json
cp ${pre}_pmc.json profiles/${tag}_cfg${cfg}_pmc.json
PT_COMMIT=$commit bash tools/pmc_cfg.sh $cfg ${tag}_cfg$cfg "$@" > ${pre}_counters.log 2>&1 && cp gpurun_out/pmc_${tag}_cfg$cfg/summary.json ${pre}_counters.json && cp ${pre}_counters.json profiles/${tag}_cfg${cfg}_counters.json
python3 tools/make_manifest.py $tag $commit > /dev/null      # so that the bench lines below name the commit and are not `stale` (the file hashes are rewritten at the end)
python3 bench.py --config $cfg "$@" > ${pre}_bench.json 2>> ${pre}.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$cfg -- python3 bench.py --config $cfg --no-cpu-baseline --no-leaves-compare "$@" > ${pre}_bench_profiled.json 2>> ${pre}.err
[ $cfg = 1 ] && python3 bench.py --config 1 --no-cpu-baseline --no-leaves-compare --overlap 0 "$@" | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); d['cpu_baseline'] = json.load(open('${pre}_bench.json'))['cpu_baseline']; print(json.dumps(d))" > ${pre}_bench_one_stream.json
cp $(find $out/kt$cfg -name "*kernel_stats.csv" | head -1) ${pre}_kernel_stats.csv
python3 tools/per_bounce.py $(find $out/kt$cfg -name "*kernel_trace.csv" | head -1) > ${pre}_per_bounce.json || true
rm -rf $out/kt$cfg $out/pf$cfg $out/pw$cfg
cat ${pre}_bench.json; head -12 ${pre}_kernel_stats.csv
