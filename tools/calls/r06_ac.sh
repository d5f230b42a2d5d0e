#!/bin/bash
# round 6, call AC: every DCN tile shape per layer shape at 4 and 8 streams on the final kernels (one launch per layer, offsets from a map)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ac; mkdir -p $O
V=64/1,128/1,32x64/1,32x64/2,32x128/1,4x32x64/1,4x32x128/1,P32x64/1,P32x64/2,P32x128/1,P4x32x64/1
for b in 4 8; do timeout 600 python tools/kbench.py --batch $b --no-conv --dvariant $V > $O/kbench_dcn_b$b.txt 2>&1; cat $O/kbench_dcn_b$b.txt; done
