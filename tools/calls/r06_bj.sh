#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_hip_plans.py tests/test_hip_fullsize.py -q 2>&1 | tail -2
