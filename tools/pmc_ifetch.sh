#!/bin/bash
# NOT RUN in round 5 (the GPU budget ended with the probe that found the separator): the confirmation step of DESIGN.md section 4.
# Instruction-cache counters of the headline frame, per kernel, in WHICHEVER state this lease is in (the state is printed and
# goes into the file names): SQ_IFETCH / SQ_IFETCH_LEVEL (fetches and their summed latency), SQC_ICACHE_REQ / _MISSES /
# _MISSES_DUPLICATE.  Expected if the slow state is a slower instruction-fetch path: the same requests and misses per launch in both
# states (the code does not change), SQ_IFETCH_LEVEL / SQ_IFETCH ~1.5x higher in the slow one, most for stem / decode stage 2.
# Counters only with --kernel-trace, each group in its own pass (gpurun refuses --pmc beside the other trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; TAG=${1:-x}; mkdir -p gpurun_out/r05_ifetch; O=$R/gpurun_out/r05_ifetch
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench.json 2>$O/${TAG}_bench.err
STATE=$(python -c "
import json
j=json.loads([l for l in open('$O/${TAG}_bench.json') if l.startswith('{')][-1])
print('slow' if j['device_ms_per_frame_batch'] > 1.03 else 'fast', j['device_ms_per_frame_batch'], j['box_calibration']['launch_us'])")
echo "$TAG $STATE"
S=$(echo $STATE | cut -d' ' -f1)
BENCH2="python $R/bench.py --steps 1 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
cd /tmp && export TMPDIR=/tmp
for pass in "A SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES" "B SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "C SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  set -- $pass; P=$1; shift
  rm -rf /tmp/pmc$P
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc$P -o pmc$P -- $BENCH2 > /dev/null 2>&1
  python $R/tools/pmc_stats.py $(ls /tmp/pmc$P/*counter_collection.csv /tmp/pmc$P/*/*counter_collection.csv 2>/dev/null | head -1) 30 > $O/${TAG}_${S}_pmc_$P.txt 2>&1
done
head -12 $O/${TAG}_${S}_pmc_A.txt | cut -c1-160
