#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_be; mkdir -p $O
S=$(date +%s); python bench.py > $O/bench.json 2> $O/bench.err; echo "wall $(( $(date +%s) - S )) s"
python -c "
import json
j=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_conv']['frac'], [ (c.get('value'), c.get('roofline',{}).get('frac')) for c in j.get('configs',[])])
print(j['cpu_baseline']['value'], j['box_calibration']['state']['instruction_fetch'])"
