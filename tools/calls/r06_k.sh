#!/bin/bash
# round 6, call K: conv / K-split buffer addressing parity, then the kernel tables of the headline frame and of coco_512 x 4
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_k; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
export CENTERTRACK_TUNE_CACHE=/tmp/tune_k.json
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_b1.json 2> $O/bench_b1.err
python bench.py --config coco_512 --streams 4 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_coco4.json 2> $O/bench_coco4.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $BENCH > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_kt/*/*.db | head -1) 40 > $O/kstats_mot17_512_b1.txt
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $BENCH --config coco_512 --streams 4 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_kt/*/*.db | head -1) 40 > $O/kstats_coco_512_b4.txt
rm -rf /tmp/prof_csv; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_csv -o kt -- $BENCH > /dev/null 2>&1
python $R/tools/rounds.py $(ls /tmp/prof_csv/*kernel_trace.csv /tmp/prof_csv/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/rounds_b1.txt 2>&1
cd $R
python - <<'PY'
import json
for f in ('bench_b1','bench_coco4'):
    try:
        j=json.loads([l for l in open('gpurun_out/r06_k/%s.json'%f) if l.startswith('{')][-1])
        print(f, j['value'], j.get('device_ms_per_frame_batch'), j.get('launches_per_frame'), j['roofline'].get('frac'), j['roofline'].get('total_ms'), j.get('roofline_conv',{}).get('frac'), j.get('roofline_conv',{}).get('total_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
cut -c1-150 $O/kstats_mot17_512_b1.txt | head -34
