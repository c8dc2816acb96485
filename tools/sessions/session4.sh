#!/bin/bash
# GPU session 4: quantised variant with per-axis delta (parity, speed), new host tests
set -o pipefail
out=gpurun_out/s4; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_golden.py tests/test_controller_host.py tests/test_node_host.py -m gpu -x -q -s > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; grep -h "grazing rays" $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['upload_ms_rank0'])"; }
for i in 1 2; do
run cfg3_q5_$i --config 3 &&
PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_w6.so run cfg3_q6_$i --config 3 &&
run cfg3_exact_$i --config 3 --traversal global_exact || exit 1
done
run cfg2_q --config 2 --steps 4 --traversal global && run cfg2_exact --config 2 --steps 4 --traversal global_exact
