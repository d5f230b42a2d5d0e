#!/bin/bash
# Where the waves of every kernel of the one-stream frame spend their issue cycles: three separate --pmc passes (LDS, VALU /
# VMEM, waits) of the headline bench command, summarised per kernel into gpurun_out/profiles_new/<tag>_pmc_issue_{lds,valu,wait}.txt
# usage (GPU box, repo root): bash tools/pmc_issue.sh <tag>
set -u
TAG=${1:-r04_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
export CENTERTRACK_TUNE_CACHE=${CENTERTRACK_TUNE_CACHE:-/tmp/tune_profile.json}
BENCH="python $R/bench.py --steps 1 --warmup 1 --frames-per-step 24 --no-cpu-baseline --no-roofline --no-resident"
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "valu:SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "wait:SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  name=${spec%%:*}; set=${spec#*:}
  rm -rf /tmp/pi_$name
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pi_$name -o pi -- $BENCH > /dev/null 2>&1
  f=$(ls /tmp/pi_$name/*counter_collection.csv /tmp/pi_$name/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_stats.py $f 24 > $OUT/${TAG}_pmc_issue_$name.txt; else echo "no csv ($name)"; fi
done
cut -c1-200 $OUT/${TAG}_pmc_issue_valu.txt | head -16
