#!/bin/bash
# re-tune the DCN schedule knobs of BASELINE's per-GPU batches (configs 3-5) on the current kernels; A/B against the pinned ones
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_i; O=$R/gpurun_out/r05_i
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print(box_calib.node().get('kernel'))"
CENTERTRACK_TUNE_VERBOSE=1 python tools/retune_dcn.py $O/tune_dcn.json 4,512,512 4,448,800 8,384,1280 1,512,512 2>&1 | grep -v amdgpu.ids | tail -30 > $O/retune.log; tail -6 $O/retune.log
python - <<'P'
import json
new = json.load(open('gpurun_out/r05_i/tune_dcn.json')); old = json.load(open('centertrack_amd/tune_table.json'))
for k in sorted(new):
    if k.startswith('dcnplan'):
        k3 = k.replace('dcnplan4', 'dcnplan3')
        print(k, new[k], 'pinned', old.get(k, old.get(k3)))
P
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes --no-resident"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', j['value'], 'dev', j['device_ms_per_frame_batch'], 'dcn', j['roofline']['total_ms'], j['roofline']['frac'], 'conv', j['roofline_conv']['total_ms'], j['roofline_conv']['frac'], j['plan_hash'])"; }
python - <<'P'
import json
new = json.load(open('gpurun_out/r05_i/tune_dcn.json'))
json.dump(new, open('gpurun_out/r05_i/table_new.json', 'w'))
P
for cfg in "coco_512 4" "nusc_800x448 4" "kitti_1280x384 4"; do
  set -- $cfg
  for rep in 1 2; do
    $B --config $1 --streams $2 2>/dev/null | show "$1 pinned"
    CENTERTRACK_TUNE_TABLE=$O/table_new.json $B --config $1 --streams $2 2>/dev/null | show "$1 retuned"
  done
done
