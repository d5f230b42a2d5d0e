#!/bin/bash
# round 6, call J: Winograd kernel diet (buffer staging with hardware zero-fill, SGPR weight offsets, staging without divisions):
# parity (whole GPU suite), same-box A/B of the conv layers against the round-5 kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_j; mkdir -p $O
V=$R/centertrack_amd/build/variants
timeout 600 python -m pytest tests/test_hip_ops.py -x -q > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
for B in 8 4 1; do
  CENTERTRACK_LIB=$V/libcentertrack_hip_r5wino.so python tools/kbench.py --batch $B --no-dcn --layers "3x3" > $O/kb_old_b${B}.txt 2>&1
  python tools/kbench.py --batch $B --no-dcn --layers "3x3" > $O/kb_new_b${B}.txt 2>&1
done
for f in old_b8 new_b8 old_b4 new_b4 old_b1 new_b1; do echo "== $f"; grep "3x3\|SUM\|layer" $O/kb_$f.txt | cut -c1-50,180-330; done
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
