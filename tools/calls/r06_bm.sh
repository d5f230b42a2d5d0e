#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bm; mkdir -p $O
python bench.py > $O/r06_last_bench_n1.json 2>$O/bench.err; tail -c 600 $O/bench.err; python -c "
import json
j=json.loads([l for l in open('$O/r06_last_bench_n1.json') if l.startswith('{')][-1])
print(j['value'], j['roofline']['frac'], j['roofline']['method'][:40], j['roofline_conv']['frac'], j['cpu_baseline']['value'], [round(c['roofline']['frac'],3) for c in j['configs']])"
