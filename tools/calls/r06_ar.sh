#!/bin/bash
# resident-chunk Winograd shapes (algo 212 / 213): parity, then per-layer times at one stream and at four
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ar; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "winograd" 2>&1 | tail -3
for b in 1 4; do
python tools/kbench.py --batch $b --no-dcn --layers "3x3 " --reps 30 > $O/kbench_b$b.txt 2>&1
python - <<PY
import re
keep=('layer','l3 3x3 128','l4 3x3 256','l5 3x3 512','off 3x3 128','off 3x3 256','off 3x3 512','SUM')
for l in open('$O/kbench_b$b.txt'):
    if l.startswith(keep): print(l.rstrip()[:400])
PY
done
