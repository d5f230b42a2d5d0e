#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_av; mkdir -p $O
python tools/dbg/tie_case.py mot17_512 0 17 0 4 32 > $O/case_mot_run17.txt 2>&1; cut -c1-400 $O/case_mot_run17.txt | tail -150
