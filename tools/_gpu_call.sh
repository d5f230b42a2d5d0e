set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CENTERTRACK_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r02b.json
(time timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "dcn or flip") > gpurun_out/t1.log 2>&1; tail -8 gpurun_out/t1.log
python tools/kbench.py --no-conv --reps 20 > gpurun_out/kbench_dcn_b1.txt 2>&1; cat gpurun_out/kbench_dcn_b1.txt
(time timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_e2e.py tests/test_hip_dropin.py -m gpu -x -q) > gpurun_out/t2.log 2>&1; tail -8 gpurun_out/t2.log
CENTERTRACK_TUNE_VERBOSE=1 python bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; cat gpurun_out/bench_r02b.json; tail -3 gpurun_out/bench_r02b.err
CENTERTRACK_TUNE_VERBOSE=1 python bench.py --streams 8 --no-cpu-baseline > gpurun_out/bench_r02b_b8.json 2>> gpurun_out/bench_r02b.err; cat gpurun_out/bench_r02b_b8.json
(time timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q) > gpurun_out/t3.log 2>&1; tail -8 gpurun_out/t3.log
