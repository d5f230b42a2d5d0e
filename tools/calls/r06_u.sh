#!/bin/bash
# round 6, call U: table build / epilogue instruction diet (both DCN kernels, tap-major table): parity, then the schedules with and
# without persistent MAIN launches launch by launch, and the persistent launches' phase stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_u; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -5 $O/tests_dcn.log
for k in 0 1; do
  timeout 300 python tools/dcn_slots.py --batch 4 --knobs 0,8,2,3,0,0,$k > $O/dcn_slots_b4_p$k.txt 2>&1
done
paste -d'|' <(cut -c1-50,100-150 $O/dcn_slots_b4_p0.txt) <(cut -c118-150 $O/dcn_slots_b4_p1.txt)
timeout 300 python tools/dcn_slots.py --batch 1 > $O/dcn_slots_b1.txt 2>&1; tail -1 $O/dcn_slots_b1.txt
python tools/dcn_phases.py --batch 4 --knobs 0,8,2,3,0,0,1 > $O/dcn_phases_b4_persist.txt 2>&1
python tools/dcn_phases.py --batch 4 --knobs 0,8,2,3,0,0,0 > $O/dcn_phases_b4.txt 2>&1
grep -A3 "node_3\]" $O/dcn_phases_b4_persist.txt $O/dcn_phases_b4.txt
