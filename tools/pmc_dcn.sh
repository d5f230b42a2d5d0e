#!/bin/bash
# PMC passes of one DCN variant of tools/kbench.py (one layer shape), summarised per kernel.
# usage (GPU box, repo root): bash tools/pmc_dcn.sh "<dcn layer substring>" <variant> <tag> [extra kbench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_dcn
mkdir -p $OUT
LAYER=$1; VAR=$2; TAG=$3; shift 3
CMD="python $R/tools/kbench.py --no-conv --reps 4 --dcn-layers $LAYER --dvariant $VAR $*"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pd$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pd$i -o pd -- $CMD > $OUT/${TAG}_run$i.log 2>&1
  f=$(ls /tmp/pd$i/*counter_collection.csv /tmp/pd$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_stats.py $f 3 > $OUT/${TAG}_pass$i.txt; else echo "no csv (pass $i)"; tail -5 $OUT/${TAG}_run$i.log; fi
done
cat $OUT/${TAG}_pass*.txt | cut -c1-260
