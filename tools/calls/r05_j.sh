#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_j; O=$R/gpurun_out/r05_j
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print(box_calib.node().get('kernel'))"
timeout 900 python -m pytest tests/test_hip_sparse_heads.py -x -q 2>&1 | tail -15 | tee $O/tests.log
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes --no-resident --no-roofline"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', j['value'], 'dev', j['device_ms_per_frame_batch'], j['launches_per_frame'])"; }
$B --config kitti_1280x384 --streams 4 2>/dev/null | show kitti4_dense
$B --config kitti_1280x384 --streams 4 --sparse-heads 2>/dev/null | show kitti4_sparse
$B 2>/dev/null | show mot1_dense
$B --sparse-heads 2>/dev/null | show mot1_sparse
