#!/bin/bash
# if this lease is in the SLOW state: the HBM-traffic / L2 hit counters of the headline frame, per kernel (to compare with the
# fast-state tables profiles/r05_b_pmc_*): does the slow state move more bytes (Infinity Cache not retaining) or the same bytes slower?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; TAG=${1:-x}; mkdir -p gpurun_out/r05_slow; O=$R/gpurun_out/r05_slow
DEV=$(python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['device_ms_per_frame_batch'], j['roofline_conv']['total_ms'])")
echo "$TAG device/conv ms: $DEV"
SLOW=$(python -c "print(1 if float('$DEV'.split()[0]) > 1.03 else 0)")
if [ "$SLOW" != "1" ] && [ "${2:-}" != "force" ]; then echo "fast state: nothing collected"; exit 0; fi
BENCH2="python $R/bench.py --steps 1 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcB /tmp/pmcC /tmp/pmcD
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcB -o pmcB -- $BENCH2 > /dev/null 2>&1
python $R/tools/pmc_stats.py $(ls /tmp/pmcB/*counter_collection.csv /tmp/pmcB/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $O/${TAG}_pmc_B.txt 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmcC -o pmcC -- $BENCH2 > /dev/null 2>&1
cd $R
for p in B C; do python tools/pmc_stats.py $(ls /tmp/pmc$p/*counter_collection.csv /tmp/pmc$p/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $O/${TAG}_pmc_$p.txt 2>&1; done
head -12 $O/${TAG}_pmc_B.txt | cut -c1-160
