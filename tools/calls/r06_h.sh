#!/bin/bash
# round 6, call H: packed blend on top of the diet: parity, same-box A/B against the round-5 kernel, then the DCN schedules
# re-measured for every pinned shape (offset mode 3 = Winograd OFFSETS launch among the candidates) and the frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_h; mkdir -p $O
V=$R/centertrack_amd/build/variants
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "dcn" > $O/tests_dcn.log 2>&1; tail -3 $O/tests_dcn.log
DV="32x64/1,32x128/1,F32x64/1,4x32x64/1,4F32x64/1"
for B in 8 4 1; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_r5dcn.so python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_old_b${B}.txt 2>&1
  python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_new_b${B}.txt 2>&1
done
for f in old_b8 new_b8 old_b4 new_b4 old_b1 new_b1; do echo "== $f"; grep "dcn \|SUM" $O/kb_$f.txt | cut -c1-110; done
CENTERTRACK_TUNE_VERBOSE=1 timeout 1200 python tools/retune_dcn.py $O/tune_new.json > $O/retune.log 2>&1
grep "dcnplan4" $O/retune.log
