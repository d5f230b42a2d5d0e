#!/bin/bash
# round 6, call C: first run of the warp-specialised DCN kernel (algo 864) and of the Winograd OFFSETS launch: parity tests,
# layer-by-layer timing against the 4-wave tiles, per-launch timing of the whole DCN schedule under the new knobs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -5 $O/tests_dcn.log
DV="32x64/1,32x64/2,64/1,ws64/1,ws64/2,ws64/4"
for B in 1 4 8; do timeout 300 python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_b$B.txt 2>&1; done
for K in 0,4,2,2,0,0,0 0,4,2,3,0,0,0 0,4,2,3,0,0,1 0,8,2,3,0,0,1 0,2,2,3,0,0,1; do
  timeout 300 python tools/dcn_slots.py --batch 4 --size 512 --knobs $K > $O/slots_b4_$K.txt 2>&1; tail -1 $O/slots_b4_$K.txt
done
for K in 128,4,4,1,0,0,0 0,4,2,3,0,0,0 0,4,4,3,0,0,0 0,4,2,3,0,0,1 0,2,2,3,0,0,1 0,2,2,3,1,0,1; do
  timeout 300 python tools/dcn_slots.py --batch 1 --size 512 --knobs $K > $O/slots_b1_$K.txt 2>&1; tail -1 $O/slots_b1_$K.txt
done
for B in 1 4 8; do echo "== batch $B"; grep "dcn \|layer\|SUM" $O/kb_b$B.txt | cut -c1-140; done
