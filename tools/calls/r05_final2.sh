#!/bin/bash
# final evidence, part 2: tie report, dense (full plan) then sparse heads (reduced headline runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_tie
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print('host kernel', box_calib.node().get('kernel'))"
( time timeout 1500 python tools/tie_report.py --out gpurun_out/r05_tie/r05_tie_report.json 2>&1 | tail -12 ) 2>&1 | tail -18
( time timeout 900 python tools/tie_report.py --out gpurun_out/r05_tie/r05_tie_report_sparse_heads.json --mot-runs 12 --sparse-heads 2>&1 | tail -12 ) 2>&1 | tail -18
