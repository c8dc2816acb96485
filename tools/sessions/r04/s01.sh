#!/bin/bash
# round 4, GPU session 1: first run of the own-leaf kernels (leaves = 2) — the GPU suite, then config 1 with both leaf modes, then the
# memory variants of both kernels A/B on config 1
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s01; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -15 $out/pytest_gpu.log
echo "pytest rc $rc"
[ $rc -ne 0 ] && [ $rc -ne 1 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_cfg1.json 2> $out/bench_cfg1.err || { tail -5 $out/bench_cfg1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_s01/bench_cfg1.json'))
print('cfg1', d['value'], d['config']['leaves'], d['config']['extend_variant'], d['config']['shadow_variant'], d['kernel_ms_rank0'], 'retraced', d['verify_failed_rank0'])
print('compare', d.get('leaves_compare'))
PY
for ev in 17 7 4 6 5; do for sv in 4 6 7; do
  PTMI_OWN_EXTEND=$ev PTMI_OWN_SHADOW=$sv timeout -k 10 120 python bench.py --no-cpu-baseline --no-leaves-compare 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extend $ev shadow $sv ->', d['config']['extend_variant'], d['config']['shadow_variant'], d['value'], d['kernel_ms_rank0'])" || echo "variant $ev $sv failed"
done; done 2>&1 | tee $out/variants_cfg1.txt
for cfg in 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg --steps 2 > $out/bench_cfg$cfg.json 2> $out/bench_cfg$cfg.err || { tail -5 $out/bench_cfg$cfg.err; continue; }
  python -c "
import json
d = json.load(open('gpurun_out/r04_s01/bench_cfg$cfg.json'))
print('cfg$cfg', d['value'], d['config']['leaves'], d['config']['extend_variant'], d['config']['shadow_variant'], d['kernel_ms_rank0'], 'retraced', d['verify_failed_rank0'], 'upload', d['upload_ms_rank0'])
print('compare', d.get('leaves_compare'))"
done
