#!/bin/bash
# round 4, GPU session 2: why config 1 did not gain from the own leaves — lane statistics, one-stream kernel times and VALU counters of
# both leaf modes, after the scalar-register fix (scene description read from memory: two workgroups per CU again)
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s02; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_own_leaves.py tests/test_gpu_parity.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for lv in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-leaves-compare --leaves $lv > $out/bench_cfg1_leaves$lv.json 2>/dev/null
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-leaves-compare --leaves $lv --overlap 0 > $out/bench_cfg1_leaves${lv}_one_stream.json 2>/dev/null
  python -c "
import json
for f in ('bench_cfg1_leaves$lv.json', 'bench_cfg1_leaves${lv}_one_stream.json'):
    d = json.load(open('gpurun_out/r04_s02/' + f)); print(f, d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"
done
for ev in 17 4; do
  PTMI_OWN_EXTEND=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --no-leaves-compare --overlap 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream, extend $ev ->', d['config']['extend_variant'], d['config']['shadow_variant'], d['value'], d['kernel_ms_rank0'])"
done
export PTMI_LIB=$root/wgpu-path-tracing_amd/lib/ab/libptmi_util.so
PTMI_OPTS='{"leaves": 1}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_stats_cfg1_leaves1.json
PTMI_OPTS='{"leaves": 2}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_stats_cfg1_leaves2.json
PTMI_OWN_EXTEND=4 PTMI_OPTS='{"leaves": 2}' timeout -k 10 200 python tools/lane_stats.py 1 > $out/lane_stats_cfg1_leaves2_extend_lds.json
unset PTMI_LIB
python - <<'PY'
import json
for f in ('lane_stats_cfg1_leaves1', 'lane_stats_cfg1_leaves2', 'lane_stats_cfg1_leaves2_extend_lds'):
    d = json.load(open(f'gpurun_out/r04_s02/{f}.json'))
    print(f, d.get('extend_variant'), d.get('shadow_variant'))
    for k in ('extend', 'shadow'):
        print('  ', k, {a: b for a, b in d[k].items() if a != 'rays'})
PY
for lv in 1 2; do
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc$lv -- python3 bench.py --no-cpu-baseline --no-leaves-compare --leaves $lv --overlap 0 --steps 1 > /dev/null 2> $out/pmc$lv.err || echo "pmc $lv failed"
  python3 tools/pmc_summary.py --json $(find $out/pmc$lv -name "*counter_collection.csv") > $out/pmc_leaves$lv.json; rm -rf $out/pmc$lv
  python3 -c "
import json
d = json.load(open('$out/pmc_leaves$lv.json'))
for k in sorted({k for c in d.values() for k in c}):
    g = lambda c: d.get(c, {}).get(k, {}).get('avg_per_launch', 0)
    if g('SQ_ACTIVE_INST_VALU'): print('leaves $lv', k, 'launches', d['SQ_INSTS_VALU'][k]['launches'], 'VALU insts/launch %.4g' % g('SQ_INSTS_VALU'), 'SALU %.4g' % g('SQ_INSTS_SALU'), 'lane util %.3f' % (g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU') / 64), 'busy ms/launch at 2.4 GHz %.3f' % (4 * g('SQ_ACTIVE_INST_VALU') / 1024 / 2.4e6))"
done
