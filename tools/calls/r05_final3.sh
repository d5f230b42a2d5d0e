#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_final3; O=$R/gpurun_out/r05_final3
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print('host kernel', box_calib.node().get('kernel'))" | tee $O/node.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time; tail -4 $O/gpu_tests.log; tail -3 $O/gpu_tests.time
