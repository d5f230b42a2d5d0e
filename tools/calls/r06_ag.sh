#!/bin/bash
# round 6, call AG: the 64-pixel x 64-cout DCN shape capped at 128 VGPRs (four workgroups per CU: ONE round of 1024 workgroups for a
# 64 -> 64 layer at four streams) against the shipped shapes, per layer shape at 4 streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ag; mkdir -p $O
V=64/1,64/2,32x64/1
timeout 300 python tools/kbench.py --batch 4 --no-conv --dvariant $V > $O/kbench_base.txt 2>&1
CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_dcn64w4.so timeout 300 python tools/kbench.py --batch 4 --no-conv --dvariant $V > $O/kbench_dcn64w4.txt 2>&1
tail -9 $O/kbench_base.txt; tail -9 $O/kbench_dcn64w4.txt
