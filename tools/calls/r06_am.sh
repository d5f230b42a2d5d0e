#!/bin/bash
# forward-only identity runs, detail of the first differing buffer; A/B against the non-Winograd OFFSETS launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_am; mkdir -p $O
timeout 300 python tools/determinism.py --model 120 --graph 0 --config coco_512 --streams 4 > $O/wino.jsonl 2> $O/wino.err
echo wino rc=$?; grep -c first $O/wino.jsonl; head -3 $O/wino.jsonl | cut -c1-3000
CENTERTRACK_DCN_KNOBS=0,8,2,1,0,0,0 timeout 300 python tools/determinism.py --model 400 --graph 0 --config coco_512 --streams 4 > $O/ksplit.jsonl 2> $O/ksplit.err
echo ksplit rc=$?; grep -c first $O/ksplit.jsonl; head -2 $O/ksplit.jsonl | cut -c1-2000
