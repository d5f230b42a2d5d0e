#!/bin/bash
# round 6, call D: ablations of the warp-specialised kernel (what holds it at 0.41 of the matrix rate?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_d; mkdir -p $O
V=$R/centertrack_amd/build/variants
DV="32x64/1,ws64/1"
python tools/kbench.py --batch 8 --no-conv --dvariant $DV --dcn-layers "64 @" > $O/kb_base.txt 2>&1
for a in 1 2 4 5; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_abl$a.so python tools/kbench.py --batch 8 --no-conv --dvariant $DV --dcn-layers "64 @" > $O/kb_abl$a.txt 2>&1
done
for f in base abl1 abl2 abl4 abl5; do echo "== $f"; grep "dcn " $O/kb_$f.txt | cut -c1-80; done
bash tools/pmc_dcn.sh "64-64" "ws64/1" b8_ws --batch 8 > $O/pmc_ws.log 2>&1
cat gpurun_out/pmc_dcn/b8_ws_pass*.txt | cut -c1-250 | grep -v elementwise
cp gpurun_out/pmc_dcn/b8_ws_pass*.txt $O/
