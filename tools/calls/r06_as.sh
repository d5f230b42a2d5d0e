#!/bin/bash
# B fragments prefetched 7 steps ahead (ring of 8) in the multi-chunk Winograd shapes: parity, per-layer times at 1 / 4 streams,
# ring of 4 (variant build) in the same lease
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_as; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "winograd or conv" 2>&1 | tail -2
for b in 1 4; do
for lib in ring8 ring4; do
L=""; [ $lib = ring4 ] && L=$R/centertrack_amd/build/variants/libcentertrack_hip_ring4.so
CENTERTRACK_LIB=$L python tools/kbench.py --batch $b --no-dcn --layers "3x3 " --reps 30 > $O/kbench_b${b}_$lib.txt 2>&1
echo "== batch $b $lib"; grep -E "^(layer|l2 3x3 64|l3 3x3 128|l4 3x3 256|l5 3x3 512|off 3x3)" $O/kbench_b${b}_$lib.txt | awk '{printf "%-22s", $1" "$2" "$3; for(i=4;i<=NF;i++) if ($i ~ /\//) printf " %s", $i; print ""}' | cut -c1-260
done; done
