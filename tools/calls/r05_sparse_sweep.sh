#!/bin/bash
# the sweep of tools/sweep_configs.sh with opt.sparse_heads (opt-in mode): one JSON line per configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/profiles_new; OUT=$R/gpurun_out/profiles_new
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print('host kernel', box_calib.node().get('kernel'))"
: > $OUT/r05_e_sweep_sparse_heads.jsonl
run() { python bench.py --config $1 --streams $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes --sparse-heads >> $OUT/r05_e_sweep_sparse_heads.jsonl 2>> $OUT/r05_e_sweep_sparse_heads.err; }
for B in 1 8 16 32; do run mot17_512 $B; done
for B in 1 4 8 16 32; do run nusc_800x448 $B; done
run kitti_1280x384 4
run coco_512 4
for B in 1 8; do run mot17_544x960 $B; done
python - <<'P'
import json
for line in open('gpurun_out/profiles_new/r05_e_sweep_sparse_heads.jsonl'):
    try: j = json.loads(line)
    except ValueError: continue
    print('%-16s x%-3d %9.1f fps  resident %9.1f  dev %.3f ms  launches %d  kernel %s' % (j['config']['workload'].split(':')[0], j['config']['global_batch'], j['value'], j.get('resident_frames_fps', 0), j['device_ms_per_frame_batch'], j['launches_per_frame'], j['box_calibration'].get('node', {}).get('kernel')))
P
