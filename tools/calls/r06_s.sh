#!/bin/bash
# round 6, call S: phase stamps of the persistent DCN launches at 4 streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_s; mkdir -p $O
python tools/dcn_phases.py --batch 4 --knobs 0,8,2,3,0,0,1 > $O/dcn_phases_b4_persist.txt 2>&1
cat $O/dcn_phases_b4_persist.txt
