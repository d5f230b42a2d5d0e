#!/bin/bash
# run-to-run identity of the frame path (tools/determinism.py): the configurations whose T = 8 plan tests failed once each
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ak; mkdir -p $O
timeout 900 python tools/determinism.py --config coco_512 --streams 4 --config kitti_1280x384 --streams 4 --config nusc_800x448 --streams 8 \
  --frames 8 --passes 5 > $O/determinism.jsonl 2> $O/determinism.err
echo rc=$?; cut -c1-600 $O/determinism.jsonl; tail -5 $O/determinism.err
