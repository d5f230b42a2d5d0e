#!/bin/bash
# Collect the round's evidence on the GPU box in ONE gpurun call:
#   bench line, rocprofv3 kernel-trace stats and the PMC passes (SQ / FETCH_SIZE / WRITE_SIZE, separate passes
#   as MI355X_MICROARCH.md prescribes) of the SAME bench command, summarised into gpurun_out/profiles_new/.
# usage (from the repo root on the GPU box):  bash tools/collect_profiles.sh <tag>     e.g. r02_a
set -u
TAG=${1:-r03_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
export CENTERTRACK_TUNE_CACHE=${CENTERTRACK_TUNE_CACHE:-/tmp/tune_profile.json}      # the profiled runs replay the tuned choices
cd $R
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/pmcA /tmp/pmcB /tmp/pmcC
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $BENCH > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_kt/*/*.db | head -1) 40 > $OUT/${TAG}_rocprofv3_kernel_stats_bench_steps3.txt
# the same run once more as a CSV trace: grid, registers and LDS of every dispatch -> whole-round table of one frame
rm -rf /tmp/prof_csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_csv -o kt -- $BENCH > /dev/null 2>&1
python $R/tools/rounds.py $(ls /tmp/prof_csv/*kernel_trace.csv /tmp/prof_csv/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/${TAG}_rounds_b1.txt 2>&1
BENCH2="python $R/bench.py --steps 1 --warmup 1 --frames-per-step 24 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcA -o pmcA -- $BENCH2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcB -o pmcB -- $BENCH2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmcC -o pmcC -- $BENCH2 > /dev/null 2>&1
cd $R
python tools/pmc_stats.py $(ls /tmp/pmcA/*counter_collection.csv /tmp/pmcA/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $OUT/${TAG}_pmc_sq.txt
python tools/pmc_stats.py $(ls /tmp/pmcB/*counter_collection.csv /tmp/pmcB/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $OUT/${TAG}_pmc_fetch_size.txt
python tools/pmc_stats.py $(ls /tmp/pmcC/*counter_collection.csv /tmp/pmcC/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $OUT/${TAG}_pmc_write_size.txt
python tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch_size.txt $OUT/${TAG}_pmc_write_size.txt $TAG > $OUT/pmc_traffic.json
python tools/pmc_busy.py $OUT/${TAG}_pmc_sq.txt $TAG > $OUT/pmc_mfma_busy.json
python tools/dcn_profiled.py $OUT/${TAG}_rocprofv3_kernel_stats_bench_steps3.txt | sed "s#$OUT/#profiles/#" > $OUT/dcn_profiled.json
cat $OUT/${TAG}_bench_n1.json
tail -12 $OUT/pmc_traffic.json
