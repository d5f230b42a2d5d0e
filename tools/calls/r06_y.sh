#!/bin/bash
# round 6, call Y: XCD-aware workgroup order of the DCN MAIN launches (dcn_xcd 1 / 0): parity, timing launch by launch at 1 and 4
# streams, FETCH_SIZE / WRITE_SIZE per launch at one stream
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_y; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -2 $O/tests_dcn.log
for x in 1 0; do for b in 1 4; do
  CENTERTRACK_TUNE=dcn_xcd=$x timeout 300 python tools/dcn_slots.py --batch $b > $O/slots_b${b}_x$x.txt 2>&1
done; done
for b in 1 4; do for x in 1 0; do echo "== b$b xcd=$x"; grep "^dcn\[" $O/slots_b${b}_x$x.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'; tail -1 $O/slots_b${b}_x$x.txt; done; done
cd /tmp && export TMPDIR=/tmp
for x in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  CENTERTRACK_TUNE=dcn_xcd=$x rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/dcn_slots.py --batch 1 > /dev/null 2>&1
  python $R/tools/pmc_stats.py $(ls /tmp/pmc_$c/*counter_collection.csv /tmp/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1) 12 > $O/pmc_${c}_b1_x$x.txt
done; done
cd $R; for x in 1 0; do echo "== xcd=$x"; grep -h "dcn_" $O/pmc_FETCH_SIZE_b1_x$x.txt $O/pmc_WRITE_SIZE_b1_x$x.txt | cut -c1-150; done
