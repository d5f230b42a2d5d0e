#!/bin/bash
# round 6, call R: the persistent DCN MAIN launch: parity (bit-identical to the per-tile launch, oracle), then the 4-stream
# schedule with and without it, launch by launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_r; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -5 $O/tests_dcn.log
for k in 0 1; do
  timeout 300 python tools/dcn_slots.py --batch 4 --knobs 0,8,2,3,0,0,$k > $O/dcn_slots_b4_p$k.txt 2>&1
done
paste -d'|' <(cut -c1-50,100-150 $O/dcn_slots_b4_p0.txt) <(cut -c118-150 $O/dcn_slots_b4_p1.txt)
