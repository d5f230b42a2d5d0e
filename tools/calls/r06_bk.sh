#!/bin/bash
# the GPU suite, smoke and the default bench line on the last tree of the round
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bk; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r06_last_pytest_gpu.log 2>&1; tail -1 $O/r06_last_pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/r06_last_pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r06_last_bench_n1.json 2>$O/bench.err; python -c "
import json
j=json.loads([l for l in open('$O/r06_last_bench_n1.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_conv']['frac'], [round(c['roofline']['frac'],3) for c in j['configs']])"
