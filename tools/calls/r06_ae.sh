#!/bin/bash
# round 6, call AE: kernel arguments of the grouped launches copied by value (wide scalar loads up front): parity, schedules at 1 / 4
# streams launch by launch (twice), XCD-order bit-identity test
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ae; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for r in 1 2; do for b in 1 4; do
  timeout 300 python tools/dcn_slots.py --batch $b > $O/slots_b${b}_$r.txt 2>&1
  echo "== b$b run $r"; grep "^dcn" $O/slots_b${b}_$r.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'; tail -1 $O/slots_b${b}_$r.txt
done; done
