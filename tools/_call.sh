#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh r04_f > gpurun_out/collect_r04_f.log 2>&1
bash tools/sweep_configs.sh r04_f prof > gpurun_out/sweep_r04_f.log 2>&1
tail -20 gpurun_out/sweep_r04_f.log
