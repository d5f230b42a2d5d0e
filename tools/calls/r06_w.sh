#!/bin/bash
# round 6, call W: FINISH kernel with 32-bit row-wise indexing: parity (ops, model), 4-stream and 1-stream schedules launch by launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
timeout 300 python tools/dcn_slots.py --batch 4 > $O/slots_b4.txt 2>&1
timeout 300 python tools/dcn_slots.py --batch 1 > $O/slots_b1.txt 2>&1
grep "finish\|total" $O/slots_b4.txt | cut -c1-40,100-160; grep "finish\|total" $O/slots_b1.txt | cut -c1-40,100-160
