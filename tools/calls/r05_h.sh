#!/bin/bash
# the default bench invocation exactly as the driver runs it (timing it), then the full GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_h; O=$R/gpurun_out/r05_h
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<'P'
import json
j = json.loads([l for l in open('gpurun_out/r05_h/bench_default.json') if l.startswith('{')][-1])
print('headline', j['value'], 'dev', j['device_ms_per_frame_batch'], 'kernel', j['box_calibration'].get('node', {}).get('kernel'))
print('sparse', j.get('sparse_heads'))
for c in j.get('configs', []):
    print(c.get('workload', '')[:40], c.get('fps'), c.get('roofline', {}).get('frac'), c.get('roofline_conv', {}).get('frac'), 'sparse', (c.get('sparse_heads') or {}).get('fps'), c.get('error'))
print('line bytes', len(json.dumps(j)))
P
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time; tail -5 $O/gpu_tests.log; tail -3 $O/gpu_tests.time
