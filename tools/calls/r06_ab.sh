#!/bin/bash
# round 6, call AB: last table pass rotated over the waves: parity, 4- and 1-stream schedules launch by launch (twice each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ab; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -2 $O/tests_dcn.log
for r in 1 2; do for b in 1 4; do
  timeout 300 python tools/dcn_slots.py --batch $b > $O/slots_b${b}_$r.txt 2>&1
  echo "== b$b run $r"; grep "^dcn\[" $O/slots_b${b}_$r.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'; tail -1 $O/slots_b${b}_$r.txt
done; done
