#!/bin/bash
# tools/ab.sh <old.so> <new.so> [rounds] [bench args...]: interleaved same-box A/B of two builds of libptmi.so
old=$1; new=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do
  for tag in old new; do
    lib=$old; [ $tag = new ] && lib=$new
    PTMI_LIB=$PWD/$lib python bench.py --no-cpu-baseline --timing 3 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'])"
  done
done
