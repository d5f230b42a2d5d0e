#!/bin/bash
# round 6, call G: same-box A/B of the DCN kernel before / after the VALU diet, + instruction counters of the new one
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_g; mkdir -p $O
V=$R/centertrack_amd/build/variants
DV="32x64/1,32x64/2,32x128/1,F32x64/1,4x32x64/1,4F32x64/1"
for rep in 1 2; do
for B in 8 1; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_r5dcn.so python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_old_b${B}_$rep.txt 2>&1
  python tools/kbench.py --batch $B --no-conv --dvariant $DV > $O/kb_new_b${B}_$rep.txt 2>&1
done
done
for f in old_b8_1 new_b8_1 old_b8_2 new_b8_2 old_b1_1 new_b1_1 old_b1_2 new_b1_2; do echo "== $f"; grep "dcn \|SUM" $O/kb_$f.txt | cut -c1-125; done
bash tools/pmc_dcn.sh "64-64" "32x64/1" b8_new --batch 8 > $O/pmc.log 2>&1
cat gpurun_out/pmc_dcn/b8_new_pass*.txt | cut -c1-250 | grep -v elementwise
cp gpurun_out/pmc_dcn/b8_new_pass*.txt $O/
