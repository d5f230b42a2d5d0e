#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_f; O=$R/gpurun_out/r05_f
python tools/box_calib.py 2>/dev/null | tail -1 > $O/box.json; cat $O/box.json
timeout 900 python -m pytest tests/test_hip_sparse_heads.py -x -q 2>&1 | tail -5 | tee $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', j['value'], 'fps resident', j.get('resident_frames_fps'), 'dev', j['device_ms_per_frame_batch'], 'launches', j['launches_per_frame'], 'dcn', j['roofline']['total_ms'], 'conv', j['roofline_conv']['total_ms'], j['box_calibration'].get('clocks_under_load',{}).get('sclk_mhz'))"; }
$B 2>/dev/null | show dense
$B --sparse-heads 2>/dev/null | show sparse
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sp
rocprofv3 --kernel-trace --stats -d /tmp/prof_sp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes --sparse-heads > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_sp/*/*.db | head -1) 60 | grep -i "wino_conv_kernel<1, 2, 1, false, 8\|decode\|sparse\|TOTAL\|stem" | tee $O/kstats_sparse.txt
