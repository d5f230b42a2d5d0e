#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bg; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_e2e.py tests/test_hip_tie_policy.py tests/test_hip_rccl.py -q 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; tail -1 $O/suite.log; grep -E "^(FAILED|ERROR)" $O/suite.log | cut -c1-160
