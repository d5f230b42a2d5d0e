#!/bin/bash
# round 6, call N: buffer stores in ct_store_tile, conv / K-split staging without divisions: parity (ops, model, full-size), frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_n; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
for cfg in "mot17_512 1 10" "coco_512 4 6" "nusc_800x448 4 6" "kitti_1280x384 4 6"; do set -- $cfg
  python bench.py --config $1 --streams $2 --steps $3 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_n/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-34s fps %8.1f dev %.4f dcn %.4f (%.3f) conv %.4f (%.3f)' % (f.split('/')[-1], j['value'], j.get('device_ms_per_frame_batch'), j['roofline'].get('total_ms'), j['roofline'].get('frac'), j.get('roofline_conv',{}).get('total_ms'), j.get('roofline_conv',{}).get('frac')))
    except Exception as e: print(f, 'ERR', e)
PY
