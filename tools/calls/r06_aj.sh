#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_aj; mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_hip_plans.py -q -k "nusc_800x448_x8" 2>&1 | tail -1; done
for i in 1 2; do timeout 900 python -m pytest tests/test_hip_plans.py -q 2>&1 | tail -2; done
