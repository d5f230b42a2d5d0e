#!/bin/bash
# forward-only identity runs: which launch's result differs first (tools/determinism.py --model)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_al; mkdir -p $O
for g in 1 0; do
timeout 600 python tools/determinism.py --model 400 --graph $g --config coco_512 --streams 4 --config kitti_1280x384 --streams 4 --config nusc_800x448 --streams 8 --config mot17_512 --streams 1 \
  > $O/model_graph$g.jsonl 2> $O/model_graph$g.err
echo graph=$g rc=$?; cut -c1-1500 $O/model_graph$g.jsonl | head -12; tail -3 $O/model_graph$g.err
done
