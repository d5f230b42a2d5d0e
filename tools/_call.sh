#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_final.json'))
r=j['roofline']; c=j['roofline_conv']
print(j['value'], j['ms_per_step'], j['launches_per_frame'], j['process_group'])
print('dcn', r['frac'], r['frac_main'], r['traffic'], r['traffic_source'][:40], r.get('traffic_over_algorithmic'), r.get('traffic_over_algorithmic_with_idaup'), r.get('mfma_busy'), r.get('waves_per_simd_avg'), r['profiled']['source'], r['profiled']['frac'])
print('conv', c['frac'], c['algorithmic_tflops'], c.get('mfma_busy'), c.get('mfma_busy_backbone_3x3'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
PY
