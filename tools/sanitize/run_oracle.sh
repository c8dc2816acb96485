#!/bin/bash
# tools/sanitize/run_oracle.sh: the three oracle libraries (contract build, literal-arithmetic build, independent literal transcription)
# under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the oracle's own CPU tests; log -> profiles/r03_sanitizers_oracle.txt
set -o pipefail
cd "$(dirname "$0")/../.."
d=/tmp/osan; mkdir -p $d; log=profiles/r03_sanitizers_oracle.txt
fl="-O1 -g -std=c11 -fopenmp -ffp-contract=off -fno-fast-math -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Ioracle -shared"
{
  echo "# $(gcc --version | head -1); $(date -u +%F); flags: $fl"
  gcc $fl -mfma -DPT_STRICT=0 -o $d/libpt_oracle.so oracle/pt_oracle.c -lm && gcc $fl -DPT_STRICT=1 -o $d/libpt_oracle_strict.so oracle/pt_oracle.c -lm && gcc $fl -o $d/libpt_literal.so oracle/pt_literal.c -lm || exit 1
  echo "## python -m pytest tests/test_oracle.py tests/test_golden.py -m 'not gpu' with the sanitized libraries preloaded"
  PT_ORACLE_BUILD_DIR=$d LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle.py tests/test_golden.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  echo "exit code ${PIPESTATUS[0]}"
} > $log 2>&1
cat $log
