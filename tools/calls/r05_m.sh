#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print('host kernel', box_calib.node().get('kernel'))"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes --no-resident"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', j['value'], j['device_ms_per_frame_batch'], 'conv', j['roofline_conv']['total_ms'])"; }
for i in 1 2; do
$B 2>/dev/null | show base
CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_nbw4.so $B 2>/dev/null | show nbw4
done
$B --config mot17_512 --streams 8 2>/dev/null | show base8
CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_nbw4.so $B --config mot17_512 --streams 8 2>/dev/null | show nbw4_8
