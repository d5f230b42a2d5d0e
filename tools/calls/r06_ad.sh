#!/bin/bash
# round 6, call AD: XCD-aware order of the Winograd OFFSETS launch: parity, one-stream FETCH_SIZE per launch, schedule timing
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ad; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for b in 1 4; do timeout 300 python tools/dcn_slots.py --batch $b > $O/slots_b$b.txt 2>&1; grep "offsets\|total" $O/slots_b$b.txt | cut -c1-30,100-160; done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_f
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/tools/dcn_slots.py --batch 1 > /dev/null 2>&1
python $R/tools/pmc_stats.py $(ls /tmp/pmc_f/*counter_collection.csv /tmp/pmc_f/*/*counter_collection.csv 2>/dev/null | head -1) 12 > $O/pmc_fetch_b1.txt; grep "dcn_\|wino_off" $O/pmc_fetch_b1.txt | cut -c1-150
