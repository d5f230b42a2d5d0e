#!/bin/bash
# round 6, call O: every launch shape of the benchmarked / tested full-size plans timed again on the round-6 kernels (pinned table
# ignored), conv and DCN schedule keys -> gpurun_out/r06_o/tune_all.json (merged into centertrack_amd/tune_table.json afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_o; mkdir -p $O
rm -f $O/tune_all.json
CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_TUNE_CACHE=$O/tune_all.json timeout 1700 python tools/tune_plans.py > $O/tune_plans.log 2>&1
tail -20 $O/tune_plans.log
python -c "
import json; t=json.load(open('$O/tune_all.json')); print(len(t), 'keys'); print({k:v for k,v in t.items() if k.startswith('dcnplan')})"
