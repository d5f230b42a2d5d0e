#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_l; O=$R/gpurun_out/r05_l
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print('host kernel', box_calib.node().get('kernel'))"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes --no-roofline --no-resident"
for m in "" "--sparse-heads" "" "--sparse-heads"; do $B $m 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('[$m]', j['value'], j['device_ms_per_frame_batch'])"; done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_l
rocprofv3 --kernel-trace --stats -d /tmp/prof_l -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes --sparse-heads > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_l/*/*.db | head -1) 60 > $O/kstats_sparse_b1.txt
grep -i "wino_conv_kernel<1, 2, 1, false, 8\|decode\|sparse\|TOTAL" $O/kstats_sparse_b1.txt
