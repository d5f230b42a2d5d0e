#!/bin/bash
# round 5, call A: the new bench line (extra configs + box probes) and "would this box pick other shapes?"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_a; O=gpurun_out/r05_a
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
( time python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -3 $O/bench_n1.time
python - <<'P'
import json
j = json.loads([l for l in open('gpurun_out/r05_a/bench_n1.json') if l.startswith('{')][-1])
print('headline', j['value'], 'fps  device', j['device_ms_per_frame_batch'], 'ms  dcn frac', j['roofline']['frac'], 'conv frac', j['roofline_conv']['frac'])
print(json.dumps(j['box_calibration'], indent=1))
for c in j.get('configs', []):
    print(c)
P
# tuner on THIS box, pinned table ignored
CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_TUNE_CACHE=$R/$O/tune_box.json python tools/tune_plans.py mot17_512:1 > $O/tune_box.log 2>&1; tail -2 $O/tune_box.log
python - <<'P'
import json
box = json.load(open('gpurun_out/r05_a/tune_box.json'))
pin = json.load(open('centertrack_amd/tune_table.json'))
same = diff = 0
for k, v in sorted(box.items()):
    if k in pin:
        if list(pin[k][:2]) == list(v[:2]):
            same += 1
        else:
            diff += 1
            print('DIFF %-60s pinned %s  box %s' % (k, pin[k], v))
print('same', same, 'different', diff, 'box-only', len([k for k in box if k not in pin]))
P
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes"
for i in 1 2; do
  $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('pinned   ', j['value'], j['device_ms_per_frame_batch'], j['roofline']['total_ms'], j['roofline_conv']['total_ms'], j['plan_hash'])"
  CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_TUNE_CACHE=$R/$O/tune_box.json $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('box-tuned', j['value'], j['device_ms_per_frame_batch'], j['roofline']['total_ms'], j['roofline_conv']['total_ms'], j['plan_hash'])"
done
