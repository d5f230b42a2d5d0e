#!/bin/bash
# round 3, call 6: the whole GPU suite on the final code + the tie report
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/r03_call6_tests.log 2>&1
tail -6 gpurun_out/r03_call6_tests.log
timeout 900 python tools/tie_report.py --out gpurun_out/r03_tie_report.json > gpurun_out/r03_tie_report.log 2>&1
tail -60 gpurun_out/r03_tie_report.log
