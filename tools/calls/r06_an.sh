#!/bin/bash
# Winograd OFFSETS race: event counts of three experiment builds (no register cap / barrier in the transpose / no preloaded scale)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_an; mkdir -p $O
for v in NOCAP SYNC NOPRE; do
CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_wx_$v.so timeout 300 python tools/determinism.py --model 300 --graph 0 --config coco_512 --streams 4 > $O/$v.jsonl 2> $O/$v.err
echo $v rc=$? events=$(grep -c first $O/$v.jsonl); tail -1 $O/$v.jsonl | cut -c1-120
done
