#!/bin/bash
# full GPU suite on the final tree (the id policy of tests/_parity.py changed after final collection 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ay; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_fin2_pytest_gpu.log 2>&1; tail -4 $O/r06_fin2_pytest_gpu.log
python bench.py > $O/r06_fin2_bench_n1.json 2> $O/bench.err; python -c "
import json
j=json.loads([l for l in open('$O/r06_fin2_bench_n1.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('box_calibration',{}).get('state',{}).get('instruction_fetch'))"
