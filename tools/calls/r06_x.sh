#!/bin/bash
# round 6, call X: the four bench lines on the table / epilogue diet + row-wise FINISH kernel (pinned table: no persistent launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_x; mkdir -p $O
for cfg in "mot17_512 1 10" "coco_512 4 6" "nusc_800x448 4 6" "kitti_1280x384 4 6"; do set -- $cfg
  python bench.py --config $1 --streams $2 --steps $3 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_x/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-34s fps %8.1f dev %.4f dcn %.4f (%.3f) conv %.4f (%.3f)' % (f.split('/')[-1], j['value'], j.get('device_ms_per_frame_batch'), j['roofline'].get('total_ms'), j['roofline'].get('frac'), j.get('roofline_conv',{}).get('total_ms'), j.get('roofline_conv',{}).get('frac')))
    except Exception as e: print(f, 'ERR', e)
PY
