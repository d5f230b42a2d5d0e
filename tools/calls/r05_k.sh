#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $R/bench.py --config mot17_512 --streams 32 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes --sparse-heads > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_k/*/*.db | head -1) 40 | grep -i "wino_conv_kernel<1, 2, 1, false, 8\|decode\|sparse\|TOTAL" 
