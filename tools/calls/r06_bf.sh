#!/bin/bash
# plan buffers staggered inside their allocations (CENTERTRACK_ALLOC_STAGGER): DCN / conv sequence times per process, 2 processes per setting
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bf; mkdir -p $O
for st in 0 256 4352 20736 0 4352; do
for cfg in "coco_512 4" "nusc_800x448 4" "mot17_512 1"; do
set -- $cfg
CENTERTRACK_ALLOC_STAGGER=$st python bench.py --config $1 --streams $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stagger %6d  %-14s x%d  fps %7.1f  device_ms %.3f  dcn_ms %.4f (frac %.3f)  conv_ms %.4f (frac %.3f)' % ($st, '$1', $2, j['value'], j.get('device_ms_per_frame_batch',0), j['roofline']['total_ms'], j['roofline']['frac'], j['roofline_conv']['total_ms'], j['roofline_conv']['frac']))" | tee -a $O/stagger.txt
done; done
