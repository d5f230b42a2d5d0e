#!/bin/bash
# counter passes at BASELINE's own per-GPU batches (coco_512 x 4, nusc_800x448 x 4): MFMA busy / waits, FETCH_SIZE, WRITE_SIZE per kernel
# (separate --pmc passes of the same bench command, as for the headline in tools/collect_profiles.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_ax; mkdir -p $O
for cfg in coco_512 nusc_800x448; do
BENCH2="python $R/bench.py --config $cfg --streams 4 --steps 1 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pA /tmp/pB /tmp/pC
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pA -o pA -- $BENCH2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pB -o pB -- $BENCH2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pC -o pC -- $BENCH2 > /dev/null 2>&1
cd $R
for p in A:sq B:fetch_size C:write_size; do
  python tools/pmc_stats.py $(ls /tmp/p${p%%:*}/*counter_collection.csv /tmp/p${p%%:*}/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $O/r06_fin_pmc_${cfg}_b4_${p##*:}.txt
done
python tools/pmc_busy.py $O/r06_fin_pmc_${cfg}_b4_sq.txt r06_fin_${cfg}_b4 > $O/pmc_mfma_busy_${cfg}_b4.json 2>$O/busy_$cfg.err
python tools/pmc_traffic.py $O/r06_fin_pmc_${cfg}_b4_fetch_size.txt $O/r06_fin_pmc_${cfg}_b4_write_size.txt r06_fin_${cfg}_b4 > $O/pmc_traffic_${cfg}_b4.json 2>$O/traffic_$cfg.err
head -14 $O/r06_fin_pmc_${cfg}_b4_sq.txt | cut -c1-250
done
