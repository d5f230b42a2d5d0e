#!/bin/bash
# one frame batch launch by launch (workgroups, registers, LDS, workgroups resident per CU, dispatch rounds): headline, coco x 4, nusc x 4
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_bh; mkdir -p $O
for c in "mot17_512 1" "coco_512 4" "nusc_800x448 4"; do
set -- $c
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_csv -o kt -- python $R/bench.py --config $1 --streams $2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes > /dev/null 2>&1
cd $R; python tools/rounds.py $(ls /tmp/prof_csv/*kernel_trace.csv /tmp/prof_csv/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/r06_fin_rounds_${1}_b$2.txt 2>&1
tail -2 $O/r06_fin_rounds_${1}_b$2.txt | cut -c1-200
done
