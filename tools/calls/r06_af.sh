#!/bin/bash
# round 6, call AF: phase stamps of the per-tile DCN kernel at 4 streams, first round of workgroups against the later ones
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_af; mkdir -p $O
python tools/dcn_phases.py --batch 4 > $O/dcn_phases_b4_rounds.txt 2>&1
grep -B1 -A4 "node_3\]\|node_1 + dla_up.ida_1.node_1" $O/dcn_phases_b4_rounds.txt | cut -c1-330
