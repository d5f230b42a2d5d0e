#!/bin/bash
# round 6, call L: Winograd epilogue through buffer stores: parity + same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_l; mkdir -p $O
V=$R/centertrack_amd/build/variants
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
for B in 4 1; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_preepi.so python tools/kbench.py --batch $B --no-dcn --layers "3x3 " > $O/kb_old_b${B}.txt 2>&1
  python tools/kbench.py --batch $B --no-dcn --layers "3x3 " > $O/kb_new_b${B}.txt 2>&1
done
for f in old_b4 new_b4 old_b1 new_b1; do echo "== $f"; grep "3x3 \|SUM\|^layer" $O/kb_$f.txt | cut -c1-50,168-330; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_b1.json 2> $O/bench_b1.err
CENTERTRACK_LIB=$V/libcentertrack_hip_preepi.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_b1_old.json 2> $O/bench_b1_old.err
python - <<'PY'
import json
for f in ('bench_b1','bench_b1_old'):
    try:
        j=json.loads([l for l in open('gpurun_out/r06_l/%s.json'%f) if l.startswith('{')][-1])
        print(f, j['value'], j.get('device_ms_per_frame_batch'), j.get('launches_per_frame'), j['roofline'].get('frac'), j['roofline'].get('total_ms'), j.get('roofline_conv',{}).get('frac'), j.get('roofline_conv',{}).get('total_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
