#!/bin/bash
# identity runs of kernel instantiations the pinned plans do not use: built-in heuristics instead of the tune table, other DCN schedule
# knobs (64-channel steps, K-split OFFSETS launch, fused offsets everywhere, persistent MAIN launches), the opt-in sparse heads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_aw; mkdir -p $O
run() { echo "== $1" | tee -a $O/variants.txt; shift; env "$@" 2>>$O/err.log | grep -o '"config": "[a-z0-9_x]*", "streams": [0-9]*, "graph": [a-z]*, "runs": [0-9]*, "events": [0-9]*' | tee -a $O/variants.txt; }
T="timeout 900 python tools/determinism.py --graph 1"
run "heuristics (CENTERTRACK_AUTOTUNE=0)" CENTERTRACK_AUTOTUNE=0 $T --model 150 --config coco_512 --streams 4 --config mot17_512 --streams 1 --config nusc_800x448 --streams 2 --config mot17_544x960 --streams 1
run "knobs 0,8,4,3,0,0,0 (64-channel steps)" CENTERTRACK_DCN_KNOBS=0,8,4,3,0,0,0 $T --model 150 --config coco_512 --streams 4 --config mot17_512 --streams 1
run "knobs 128,4,2,1,0,0,0 (K-split OFFSETS, fused below 128)" CENTERTRACK_DCN_KNOBS=128,4,2,1,0,0,0 $T --model 150 --config coco_512 --streams 4 --config mot17_512 --streams 1
run "knobs 256,2,2,1,1,1,0 (everything fused, fine splits)" CENTERTRACK_DCN_KNOBS=256,2,2,1,1,1,0 $T --model 150 --config mot17_512 --streams 1 --config kitti_1280x384 --streams 1
run "knobs 0,8,2,3,0,0,1 (persistent MAIN launches)" CENTERTRACK_DCN_KNOBS=0,8,2,3,0,0,1 $T --model 150 --config coco_512 --streams 4 --config mot17_512 --streams 8
run "knobs 0,4,2,2,0,0,0 (Winograd offset conv per layer)" CENTERTRACK_DCN_KNOBS=0,4,2,2,0,0,0 $T --model 150 --config coco_512 --streams 4
run "knobs 0,4,2,0,0,0,0 (direct offset conv per layer)" CENTERTRACK_DCN_KNOBS=0,4,2,0,0,0,0 $T --model 150 --config coco_512 --streams 4
echo "== streams with sparse heads" | tee -a $O/variants.txt
timeout 600 python tools/determinism.py --sparse-heads --config coco_512 --streams 4 --config nusc_800x448 --streams 4 --frames 8 --passes 3 2>>$O/err.log | cut -c1-160 | tee -a $O/variants.txt
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
