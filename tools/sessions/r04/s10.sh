#!/bin/bash
# round 4, GPU session 10: loose ends — shade's grid size beside the new traversal kernels, 128-frame batches, cornell_spheres with its
# nodes in memory instead of LDS (6 instead of 4 waves per SIMD), the multi handle with one enqueuing thread per device
set -o pipefail
root=$PWD; out=$root/gpurun_out/r04_s10; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $root
timeout -k 10 300 python -m pytest tests/test_gpu_multi.py tests/test_multi_rank_gpu.py -m gpu -x -q > $out/pytest_multi.log 2>&1 || { tail -30 $out/pytest_multi.log; exit 1; }
tail -2 $out/pytest_multi.log
b() { python bench.py --no-cpu-baseline --no-leaves-compare "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0']; print('$TAG', d['value'], d['config']['extend_variant'], d['config']['shadow_variant'], 'extend %.2f shade %.2f shadow %.2f' % (k['extend'], k['shade'], k['shadow']))"; }
ab=$root/wgpu-path-tracing_amd/lib/ab
for round in 1 2; do
  TAG="base (16 shade wgs per CU)" b
  for v in swg8 swg24 swg32; do TAG="$v" PTMI_LIB=$ab/libptmi_$v.so b; done
  TAG="128-frame batches" b --frames-per-step 128 --frames-per-batch 128 --steps 1
  TAG="2 steps of 64    " b --steps 2
done 2>&1 | tee $out/ab_cfg1.txt
for round in 1 2; do
  TAG="cfg2 lds / lds      " b --config 2 --steps 2
  TAG="cfg2 memory / memory" PTMI_OWN_EXTEND=8 PTMI_OWN_SHADOW=8 b --config 2 --steps 2
  TAG="cfg2 lds / memory   " PTMI_OWN_SHADOW=8 b --config 2 --steps 2
  TAG="cfg2 memory / lds   " PTMI_OWN_EXTEND=8 b --config 2 --steps 2
done 2>&1 | tee $out/ab_cfg2.txt
for n in 2 8; do timeout -k 10 200 python bench.py --gpus $n --single-process --rehearse --config 4 --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single process N=$n', d['value'], d['enqueue_ms_per_step'], d['rehearsal'])"; done 2>&1 | tee $out/multi_threads.txt
