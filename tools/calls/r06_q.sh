#!/bin/bash
# round 6, call Q: where a DCN workgroup's life goes at 4 and 1 streams (s_memtime stamps, debug library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_q; mkdir -p $O
python tools/dcn_phases.py --batch 4 > $O/dcn_phases_b4.txt 2>&1
python tools/dcn_phases.py --batch 1 > $O/dcn_phases_b1.txt 2>&1
cat $O/dcn_phases_b4.txt
