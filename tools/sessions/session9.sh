#!/bin/bash
# GPU session 9: the committed evidence of round 2 — per config: un-profiled line, kernel stats, FETCH/WRITE passes; counters
set -o pipefail
for c in 1 3 2 4; do timeout -k 10 600 bash tools/profile_round.sh r02 $c 0eeb90c > gpurun_out/prof_r02_cfg$c.log 2>&1 || { tail -5 gpurun_out/prof_r02_cfg$c.log; exit 1; }; tail -3 gpurun_out/prof_r02_cfg$c.log | cut -c1-400; done
bash tools/pmc_cfg.sh 1 r02_cfg1 > gpurun_out/pmc_r02_cfg1.log 2>&1; cat gpurun_out/pmc_r02_cfg1.log
bash tools/pmc_cfg.sh 3 r02_cfg3 > gpurun_out/pmc_r02_cfg3.log 2>&1; cat gpurun_out/pmc_r02_cfg3.log
