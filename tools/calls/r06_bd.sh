#!/bin/bash
# flake check of the GPU suite: three runs in a row on one lease, failures listed
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bd; mkdir -p $O
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q > $O/suite_$i.log 2>&1; tail -1 $O/suite_$i.log | tee -a $O/suite_x3.txt; grep -E "^(FAILED|ERROR)" $O/suite_$i.log | cut -c1-160; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
