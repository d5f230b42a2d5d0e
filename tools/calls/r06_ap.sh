#!/bin/bash
# identity runs over every full-size plan shape of the pinned table (other kernel instantiations than the four in the test)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ap; mkdir -p $O
run() { timeout 600 python tools/determinism.py --model $1 --graph 1 --config $2 --streams $3 2>>$O/err.log | tail -1 | grep -o '"config": "[a-z0-9_x]*", "streams": [0-9]*, "graph": [a-z]*, "runs": [0-9]*, "events": [0-9]*' | tee -a $O/shapes.txt; }
run 200 mot17_512 8; run 120 mot17_512 16; run 80 mot17_512 32
run 300 nusc_800x448 1; run 200 nusc_800x448 4; run 100 nusc_800x448 16; run 60 nusc_800x448 32
run 200 kitti_1280x384 1; run 200 kitti_1280x384 2
run 300 mot17_544x960 1; run 120 mot17_544x960 8
run 200 coco_512 2; run 200 coco_512 8
