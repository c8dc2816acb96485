#!/bin/bash
# GPU session 53: the engine clock the kernels actually run at (config 1 and 3)
set -o pipefail
mkdir -p gpurun_out/s53
bash tools/clock_under_load.sh 1 gpurun_out/s53/clock_cfg1.json && bash tools/clock_under_load.sh 3 gpurun_out/s53/clock_cfg3.json
