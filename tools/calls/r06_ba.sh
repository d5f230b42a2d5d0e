#!/bin/bash
# the GPU suite (flake check on one lease), failures listed
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ba; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite_a.log 2>&1; tail -1 $O/suite_a.log; grep -E "^(FAILED|ERROR)" $O/suite_a.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_b.log 2>&1; tail -1 $O/suite_b.log; grep -E "^(FAILED|ERROR)" $O/suite_b.log | cut -c1-200
