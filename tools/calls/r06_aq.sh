#!/bin/bash
# phase stamps of the backbone Winograd / conv launches at ONE stream (where levels 3-5 are latency chains)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_aq; mkdir -p $O
python tools/conv_phases.py --config mot17_512 --streams 1 --only wino,conv --loop > $O/conv_phases_mot_b1.txt 2>&1
cut -c1-250 $O/conv_phases_mot_b1.txt | head -150
