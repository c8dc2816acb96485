#!/bin/bash
# tools/asm.sh <file-stem> [extra flags]: device ISA of one csrc/*.hip into /tmp/<stem>.s, with per-kernel register / scratch use
stem=$1; shift
cd /root/repo/wgpu-path-tracing_amd
fl=""; [ $stem = traverse -o $stem = traverse_own ] && fl="-mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-rdc -fno-slp-vectorize -I../include -Icsrc $fl "$@" -S --cuda-device-only -o /tmp/$stem.s csrc/$stem.hip 2>&1 | grep -v "warning\|^$" | tail -3
grep "\.name:\|\.vgpr_count\|private_segment_fixed\|\.sgpr_count" /tmp/$stem.s | paste - - - - | awk '{print $2, "scratch", $4, "sgpr", $6, "vgpr", $8}' | sed 's/_ZN12_GLOBAL__N_1//'
