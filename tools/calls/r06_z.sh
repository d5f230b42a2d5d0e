#!/bin/bash
# round 6, call Z: persistent against per-tile MAIN launches for every pinned DCN schedule (tools/tune_persist.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_z; mkdir -p $O
timeout 1500 python tools/tune_persist.py $O/tune_persist.json > $O/tune_persist.log 2>&1
grep dcnplan5 $O/tune_persist.log
