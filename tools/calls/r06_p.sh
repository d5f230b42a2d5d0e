#!/bin/bash
# round 6, call P: the re-measured table (call O) in place: plan / full-size parity, the four bench lines, per-kernel tables of
# the 4-stream plans and of the headline, per-slot DCN timing at 4 streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_p; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_plans.py tests/test_hip_fullsize.py tests/test_hip_model.py -x -q > $O/tests_plans.log 2>&1; tail -3 $O/tests_plans.log
for cfg in "mot17_512 1 10" "coco_512 4 6" "nusc_800x448 4 6" "kitti_1280x384 4 6"; do set -- $cfg
  python bench.py --config $1 --streams $2 --steps $3 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
done
python tools/dcn_slots.py --batch 4 > $O/dcn_slots_b4.txt 2>&1
python tools/dcn_slots.py --batch 1 > $O/dcn_slots_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "mot17_512 1" "coco_512 4" "nusc_800x448 4"; do set -- $cfg
  rm -rf /tmp/prof_sw
  rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -- python $R/bench.py --config $1 --streams $2 --steps 3 --warmup 1 \
      --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_sw/*/*.db | head -1) 40 > $O/kstats_$1_b$2.txt
done
cd $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_p/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-34s fps %8.1f dev %.4f dcn %.4f (%.3f) conv %.4f (%.3f)' % (f.split('/')[-1], j['value'], j.get('device_ms_per_frame_batch'), j['roofline'].get('total_ms'), j['roofline'].get('frac'), j.get('roofline_conv',{}).get('total_ms'), j.get('roofline_conv',{}).get('frac')))
    except Exception as e: print(f, 'ERR', e)
PY
