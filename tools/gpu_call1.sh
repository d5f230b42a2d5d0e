#!/bin/bash
# round 3, call 1: new parity tests on the benchmarked plans, pose variants, gather hook; tie report; bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_plans.py tests/test_hip_e2e.py tests/test_hip_ops.py -k "plan or pose or gather or decode" -x -q -m gpu > gpurun_out/r03_call1_tests.log 2>&1
tail -5 gpurun_out/r03_call1_tests.log
timeout 900 python tools/tie_report.py --out gpurun_out/r03_tie_report.json > gpurun_out/r03_tie_report.log 2>&1
tail -40 gpurun_out/r03_tie_report.log
timeout 600 python bench.py > gpurun_out/r03_call1_bench.json 2> gpurun_out/r03_call1_bench.err
cat gpurun_out/r03_call1_bench.json
