#!/bin/bash
# round 6, call B: what bounds a step of the DCN MAIN loop in the matrix-bound regime (8 streams)?  Ablation builds
# (-DCT_ABL: 1 no corner loads, 2 no MFMAs, 4 no weight loads, 8 no barrier, 16 no A reads) timed layer by layer, and the SQ / TCP
# counter passes of the 64 -> 64 @ 128 x 128 layer on the shipped kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_b; mkdir -p $O
V=$R/centertrack_amd/build/variants
DV="32x64/1,4x32x64/1,64/1,32x128/1,64/2"
python tools/kbench.py --batch 8 --no-conv --dvariant $DV > $O/kb_base.txt 2>&1
for a in 1 2 4 8 16 3 6 7; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_abl$a.so python tools/kbench.py --batch 8 --no-conv --dvariant $DV > $O/kb_abl$a.txt 2>&1
done
grep -h "64-64 @128\|128-64 @64\|256-256" $O/kb_*.txt | cut -c1-120 > $O/summary.txt
for f in base abl1 abl2 abl4 abl8 abl16 abl3 abl6 abl7; do echo "== $f"; grep "dcn " $O/kb_$f.txt | cut -c1-110; done > $O/all.txt
bash tools/pmc_dcn.sh "64-64" "32x64/1" b8_3264 --batch 8 > $O/pmc_3264.log 2>&1
bash tools/pmc_dcn.sh "64-64" "4x32x64/1" b8_43264 --batch 8 > $O/pmc_43264.log 2>&1
cp -r gpurun_out/pmc_dcn $O/
cd /tmp && rocprofv3 -L > $O/counters_avail.txt 2>&1
cat $O/all.txt
