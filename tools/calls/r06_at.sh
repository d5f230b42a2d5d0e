#!/bin/bash
# a third workgroup per CU for the multi-chunk 64 x 32 Winograd shape (168 VGPRs, ONE LDS patch buffer): per-layer times at 4 streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_at; mkdir -p $O
for lib in base wm3s wm2s wm3; do
L=""; [ $lib != base ] && L=$R/centertrack_amd/build/variants/libcentertrack_hip_$lib.so
for b in 4; do
CENTERTRACK_LIB=$L python tools/kbench.py --batch $b --no-dcn --layers "3x3 " --variant w64x32 --reps 30 > $O/kbench_b${b}_$lib.txt 2>&1
echo "== batch $b $lib"; grep -E "^(l2 3x3 64|l3 3x3 128|l4 3x3 256|l5 3x3 512|off 3x3)" $O/kbench_b${b}_$lib.txt | cut -c1-60
done; done
CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_wm3s.so timeout 300 python -m pytest tests/test_hip_ops.py -q -k "winograd" 2>&1 | tail -2
