#!/bin/bash
# PMC passes (LDS / VMEM / VALU issue counters) of one conv variant of tools/kbench.py, summarised per kernel.
# usage (GPU box, repo root): bash tools/pmc_kernel.sh "<layer substring>" <variant> [extra kbench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_kernel
mkdir -p $OUT
LAYER=$1; VAR=$2; shift 2
CMD="python $R/tools/kbench.py --no-dcn --reps 10 --layers $LAYER --variant $VAR $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVE_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pk$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk$i -o pk -- $CMD > $OUT/run$i.log 2>&1
  f=$(ls /tmp/pk$i/*counter_collection.csv /tmp/pk$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_stats.py $f 6 > $OUT/pass$i.txt; else echo "no csv (pass $i)"; tail -5 $OUT/run$i.log; fi
done
cat $OUT/pass*.txt | cut -c1-250
