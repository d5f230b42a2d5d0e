set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CENTERTRACK_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r02a.json
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/gpu_tests_r02a.log 2>&1
tail -15 gpurun_out/gpu_tests_r02a.log
(time python bench.py) > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err
cat gpurun_out/bench_r02a.json; tail -3 gpurun_out/bench_r02a.err
python bench.py --streams 8 --no-cpu-baseline > gpurun_out/bench_r02a_b8.json 2>> gpurun_out/bench_r02a.err
cat gpurun_out/bench_r02a_b8.json
python bench.py --config kitti_1280x384 --streams 4 --no-cpu-baseline > gpurun_out/bench_r02a_kitti.json 2>> gpurun_out/bench_r02a.err
cat gpurun_out/bench_r02a_kitti.json
