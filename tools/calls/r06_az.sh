#!/bin/bash
# soak: 4000 forward runs per configuration (graph launches) + long streams through fresh detectors
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_az; mkdir -p $O
timeout 1500 python tools/determinism.py --model 4000 --graph 1 --config coco_512 --streams 4 --config mot17_512 --streams 1 --config nusc_800x448 --streams 4 2>$O/err.log | grep -o '"config": "[a-z0-9_x]*", "streams": [0-9]*, "graph": [a-z]*, "runs": [0-9]*, "events": [0-9]*' | tee $O/soak.txt
timeout 900 python tools/determinism.py --config mot17_512 --streams 1 --config kitti_1280x384 --streams 4 --frames 32 --passes 3 --no-heads 2>>$O/err.log | cut -c1-120 | tee -a $O/soak.txt
