#!/bin/bash
# round 6, call AI: phase stamps of the backbone / heads launches of coco_512 x 4 streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ai; mkdir -p $O
python tools/conv_phases.py --config coco_512 --streams 4 --only wino,conv --loop > $O/conv_phases_coco_b4.txt 2>&1
cut -c1-330 $O/conv_phases_coco_b4.txt | head -120
