#!/bin/bash
# One JSON line (+ optionally one rocprofv3 kernel-stats table) per configuration north_star / SURVEY 8(d) name:
#   synthetic sweep   B in {1, 8, 16, 32} streams at 512x512 (MOT heads) and 800x448 (nuScenes 3D heads)
#   config 3          kitti_1280x384, 4 streams, flip_test (8 images per step)
#   config 4          coco_512, 4 streams per GPU (32 streams over 8 GPUs)
# usage (repo root, GPU box):  bash tools/sweep_configs.sh <tag> [prof]     e.g.  bash tools/sweep_configs.sh r02_a prof
# Writes gpurun_out/profiles_new/<tag>_sweep.jsonl and, with `prof`, <tag>_kstats_<config>_b<B>.txt.
set -u
TAG=${1:-r03_x}
PROF=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
export CENTERTRACK_TUNE_CACHE=${CENTERTRACK_TUNE_CACHE:-/tmp/tune_sweep.json}
: > $OUT/${TAG}_sweep.jsonl
python $R/tools/box_calib.py > $OUT/${TAG}_box.json 2>/dev/null   # (every bench line carries its own box_calibration since round 3)
run() {  # config streams
    cd $R
    python bench.py --config $1 --streams $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes \
        >> $OUT/${TAG}_sweep.jsonl 2>> $OUT/${TAG}_sweep.err
    if [ -n "$PROF" ]; then
        cd /tmp && export TMPDIR=/tmp
        rm -rf /tmp/prof_sw
        rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -- python $R/bench.py --config $1 --streams $2 --steps 3 --warmup 1 \
            --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes > /dev/null 2>&1
        python $R/tools/rocpd_stats.py $(ls /tmp/prof_sw/*/*.db | head -1) 30 > $OUT/${TAG}_kstats_$1_b$2.txt
    fi
}
for B in 1 8 16 32; do run mot17_512 $B; done
for B in 1 4 8 16 32; do run nusc_800x448 $B; done
run kitti_1280x384 4
run coco_512 4
for B in 1 8; do run mot17_544x960 $B; done      # the reference's own MOT input size (datasets/mot.py:15)
cat $OUT/${TAG}_box.json
cat $OUT/${TAG}_sweep.jsonl | python -c "
import json, sys
for line in sys.stdin:
    try:
        j = json.loads(line)
    except ValueError:
        continue
    r = j.get('roofline', {})
    print('%-75s %9.1f fps  resident %9.1f  dev %.3f ms  dcn frac %.3f' % (j['config']['workload'][:75], j['value'], j.get('resident_frames_fps', 0), j.get('device_ms_per_frame_batch', 0), r.get('frac', 0)))
"
