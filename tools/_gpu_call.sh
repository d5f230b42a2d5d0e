cd $GRAFT_REPO_ROOT
R=$PWD; OUT=$R/gpurun_out/profiles_new; mkdir -p $OUT
BENCH2="python $R/bench.py --streams 32 --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-roofline --no-resident"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcB /tmp/pmcC
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcB -o pmcB -- $BENCH2 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcC -o pmcC -- $BENCH2 > /dev/null 2>&1
cd $R
python tools/pmc_stats.py $(ls /tmp/pmcB/*counter_collection.csv /tmp/pmcB/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $OUT/r02_g_b32_pmc_fetch_size.txt
python tools/pmc_stats.py $(ls /tmp/pmcC/*counter_collection.csv /tmp/pmcC/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $OUT/r02_g_b32_pmc_write_size.txt
python tools/pmc_traffic.py $OUT/r02_g_b32_pmc_fetch_size.txt $OUT/r02_g_b32_pmc_write_size.txt r02_g_b32 > $OUT/pmc_traffic_b32.json
tail -8 $OUT/pmc_traffic_b32.json
head -12 $OUT/r02_g_b32_pmc_fetch_size.txt | cut -c1-160
