#!/bin/bash
# GPU session 44: 1/x and sqrt(x) by short sequences (exhaustively equal to the IEEE expansions within [2^-100, 2^100]) —
# A/B against the same source built with -DPT_IEEE_EXPANSIONS=1 (the exhaustive test and the whole GPU suite passed in the first
# run of this session: 4 + 116 tests)
set -o pipefail
out=gpurun_out/s44; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc = 0 ] || exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err || exit 1; python -c "
import json; d=json.load(open('$out/$tag.json')); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"; }
ab=$PWD/wgpu-path-tracing_amd/lib/ab
for i in 1 2 3; do PTMI_LIB=$ab/libptmi_ieee.so run cfg1_ieee_$i --config 1 && run cfg1_short_$i --config 1 || exit 1; done
PTMI_LIB=$ab/libptmi_ieee.so run cfg1_ieee_one --config 1 --overlap 0 && run cfg1_short_one --config 1 --overlap 0
for i in 1 2; do PTMI_LIB=$ab/libptmi_ieee.so run cfg3_ieee_$i --config 3 && run cfg3_short_$i --config 3 || exit 1; done
PTMI_LIB=$ab/libptmi_ieee.so run cfg2_ieee --config 2 && run cfg2_short --config 2
PTMI_LIB=$ab/libptmi_ieee.so run cfg4_ieee --config 4 && run cfg4_short --config 4
