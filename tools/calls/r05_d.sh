#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_d; O=gpurun_out/r05_d
timeout 900 python -m pytest tests/test_hip_sparse_heads.py -x -q 2>&1 | tail -25 | tee $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', j['value'], 'fps resident', j.get('resident_frames_fps'), 'dev', j['device_ms_per_frame_batch'], 'launches', j['launches_per_frame'], 'd2d', j['box_calibration']['d2d_1GiB_GBps'])"; }
$B 2>$O/err1 | show dense
$B --sparse-heads 2>$O/err2 | show sparse
$B --config nusc_800x448 --streams 4 2>/dev/null | show nusc4_dense
$B --config nusc_800x448 --streams 4 --sparse-heads 2>/dev/null | show nusc4_sparse
$B --config mot17_512 --streams 32 2>/dev/null | show mot32_dense
$B --config mot17_512 --streams 32 --sparse-heads 2>/dev/null | show mot32_sparse
tail -3 $O/err2
