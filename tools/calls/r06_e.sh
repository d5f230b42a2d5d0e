#!/bin/bash
# round 6, call E: wave priorities (producers of the warp-specialised kernel above its matrix waves; the 4-wave kernel's MFMA burst
# below the rest of its step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_e; mkdir -p $O
V=$R/centertrack_amd/build/variants
DV="32x64/1,4x32x64/1,ws64/1,ws64/2"
python tools/kbench.py --batch 8 --no-conv --dvariant $DV > $O/kb_base.txt 2>&1
for a in wsp1 wsp3 kp1; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_$a.so python tools/kbench.py --batch 8 --no-conv --dvariant $DV > $O/kb_$a.txt 2>&1
done
for f in base wsp1 wsp3 kp1; do echo "== $f"; grep "dcn \|SUM" $O/kb_$f.txt | cut -c1-100; done
