#!/bin/bash
# round 6, call F: the VALU diet of the 4-wave DCN kernel (buffer addressing, cheaper table, buffer stores): parity, layers, slots, frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_f; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -3 $O/tests_dcn.log
DV="32x64/1,32x64/2,4x32x64/1,32x128/1,F32x64/1,4F32x64/1,ws64/1"
for B in 1 4 8; do timeout 300 python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_b$B.txt 2>&1; done
for B in 1 4 8; do echo "== batch $B"; grep "dcn \|SUM" $O/kb_b$B.txt | cut -c1-140; done
for K in 0,4,2,2,0,0,0 0,4,2,3,0,0,0; do
  timeout 300 python tools/dcn_slots.py --batch 4 --size 512 --knobs $K > $O/slots_b4_$K.txt 2>&1; tail -1 $O/slots_b4_$K.txt
done
for K in 128,4,4,1,0,0,0 0,4,2,3,0,0,0 64,4,4,3,0,0,0; do
  timeout 300 python tools/dcn_slots.py --batch 1 --size 512 --knobs $K > $O/slots_b1_$K.txt 2>&1; tail -1 $O/slots_b1_$K.txt
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_b1.json 2> $O/bench_b1.err
python bench.py --config coco_512 --streams 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/bench_coco4.json 2> $O/bench_coco4.err
python - <<'PY'
import json
for f in ('bench_b1','bench_coco4'):
    try:
        j=json.loads([l for l in open('gpurun_out/r06_f/%s.json'%f) if l.startswith('{')][-1])
        print(f, j['value'], j.get('device_ms_per_frame_batch'), j['roofline'].get('frac'), j['roofline'].get('total_ms'), j.get('roofline_conv',{}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
