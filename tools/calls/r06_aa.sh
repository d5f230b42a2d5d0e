#!/bin/bash
# round 6, call AA: persistent launches, finer stamps (first two steps / table build apart)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_aa; mkdir -p $O
python tools/dcn_phases.py --batch 4 --knobs 0,8,2,3,0,0,1 > $O/dcn_phases_b4_persist.txt 2>&1
grep -A2 "node_3\]\|node_2 + dla" $O/dcn_phases_b4_persist.txt
