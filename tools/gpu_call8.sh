#!/bin/bash
# round 3, call 8: the round's evidence -- bench line + rocprofv3 kernel trace + PMC passes (collect_profiles), sweep of all configurations
cd ${GRAFT_REPO_ROOT:-.}
bash tools/collect_profiles.sh r03_a > gpurun_out/r03_collect.log 2>&1
tail -30 gpurun_out/r03_collect.log
bash tools/sweep_configs.sh r03_a prof > gpurun_out/r03_sweep.log 2>&1
tail -16 gpurun_out/r03_sweep.log
