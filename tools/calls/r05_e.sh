#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_e; O=$R/gpurun_out/r05_e
timeout 900 python -m pytest tests/test_hip_sparse_heads.py -x -q 2>&1 | tail -25 | tee $O/tests.log
cd /tmp && export TMPDIR=/tmp
for mode in "" "--sparse-heads"; do
  rm -rf /tmp/prof_sp
  rocprofv3 --kernel-trace --stats -d /tmp/prof_sp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes $mode > /dev/null 2>&1
  echo "== mode [$mode]"
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_sp/*/*.db | head -1) 60 | grep -i "wino_conv_kernel<1, 2, 1, false, 8\|decode\|sparse\|TOTAL\|fill_words" | tee -a $O/kstats_sparse.txt
done
