set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CENTERTRACK_TUNE_PINNED=0
export CENTERTRACK_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r02e.json
rm -f $CENTERTRACK_TUNE_CACHE
bash tools/sweep_configs.sh r02_e prof > gpurun_out/sweep_r02e.log 2>&1; tail -13 gpurun_out/sweep_r02e.log
bash tools/collect_profiles.sh r02_e > gpurun_out/collect_r02e.log 2>&1; tail -16 gpurun_out/collect_r02e.log
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/gpu_tests_r02e.log 2>&1
tail -6 gpurun_out/gpu_tests_r02e.log
