#!/bin/bash
# after the store-hazard fix: identity runs (400 per configuration, graph + eager), the new tests, the T = 8 plan tests twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ao; mkdir -p $O
for g in 1 0; do
timeout 600 python tools/determinism.py --model 400 --graph $g --config coco_512 --streams 4 --config kitti_1280x384 --streams 4 --config nusc_800x448 --streams 8 --config mot17_512 --streams 1 \
  > $O/model_graph$g.jsonl 2> $O/model_graph$g.err
echo graph=$g rc=$?; grep -o '"config": "[a-z0-9_x]*", "streams": [0-9]*, "graph": [a-z]*, "runs": [0-9]*, "events": [0-9]*' $O/model_graph$g.jsonl
done
timeout 900 python tools/determinism.py --config coco_512 --streams 4 --config kitti_1280x384 --streams 4 --config nusc_800x448 --streams 8 --frames 8 --passes 4 > $O/streams.jsonl 2>$O/streams.err
echo streams rc=$?; cut -c1-200 $O/streams.jsonl
timeout 900 python -m pytest tests/test_hip_determinism.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 900 python -m pytest tests/test_hip_plans.py tests/test_hip_fullsize.py -q 2>&1 | tail -2; done
