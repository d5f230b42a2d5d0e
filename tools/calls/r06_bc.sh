#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bc; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_e2e.py -q -k "collection_of_an_old" 2>&1 | tail -12 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; tail -1 $O/suite.log; grep -E "^(FAILED|ERROR)" $O/suite.log | cut -c1-160
