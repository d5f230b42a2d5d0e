#!/bin/bash
# round 6, call V: persistent DCN launches with priority by progress (default) against the same without (variant build), and
# other slot counts; 4 streams, launch by launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_v; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -2 $O/tests_dcn.log
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python tools/dcn_slots.py --batch 4 --knobs 0,8,2,3,0,0,1 > $O/slots_$tag.txt 2>&1
}
timeout 300 python tools/dcn_slots.py --batch 4 --knobs 0,8,2,3,0,0,0 > $O/slots_base.txt 2>&1
run prio A=1
run noprio CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_noprio.so
run s768 CENTERTRACK_TUNE=dcn_slots=768
run s1280 CENTERTRACK_TUNE=dcn_slots=1280
run s2048 CENTERTRACK_TUNE=dcn_slots=2048
for t in base prio noprio s768 s1280 s2048; do echo "== $t"; grep "^dcn\[" $O/slots_$t.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'; tail -1 $O/slots_$t.txt; done
