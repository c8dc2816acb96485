#!/bin/bash
# round 4, GPU session 4: verification and result write deferred to the refill; any-hit kernel from two workgroups per CU by default.
# Own-leaf and parity tests, then config 1 against leaves = 1 on the same box; VALU counters of the new extend / shadow
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s04; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], d['kernel_ms_rank0'], d['verify_failed_rank0'])"; }
for round in 1 2 3; do
  TAG="own          " b
  TAG="leaves 1     " b --leaves 1
done 2>&1 | tee $out/ab_cfg1.txt
TAG="own one stream" b --overlap 0
for cfg in 2 3; do TAG="cfg$cfg own " b --config $cfg --steps 2; TAG="cfg$cfg leaves 1" b --config $cfg --steps 2 --leaves 1; done 2>&1 | tee $out/ab_cfg23.txt
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc -- python3 bench.py --no-cpu-baseline --no-leaves-compare --overlap 0 --steps 1 > /dev/null 2> $out/pmc.err || echo "pmc failed"
python3 tools/pmc_summary.py --json $(find $out/pmc -name "*counter_collection.csv") > $out/pmc_own.json; rm -rf $out/pmc
python3 -c "
import json
d = json.load(open('$out/pmc_own.json'))
for k in sorted({k for c in d.values() for k in c}):
    g = lambda c: d.get(c, {}).get(k, {}).get('avg_per_launch', 0)
    if g('SQ_ACTIVE_INST_VALU'): print(k, 'launches', d['SQ_INSTS_VALU'][k]['launches'], 'VALU insts/launch %.4g' % g('SQ_INSTS_VALU'), 'SALU %.4g' % g('SQ_INSTS_SALU'), 'lane util %.3f' % (g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU') / 64), 'busy ms/launch at 2.4 GHz %.3f' % (4 * g('SQ_ACTIVE_INST_VALU') / 1024 / 2.4e6))"
