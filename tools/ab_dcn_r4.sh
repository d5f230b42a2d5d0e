#!/bin/bash
# First GPU call of round 4: the two DCN experiments prepared (blind) at the end of round 3, as A/B pairs inside ONE call.
#   1. parity of the 16-pixel shape (algo 41664) -- tests/test_hip_experimental.py; nothing below counts if this is red
#   2. base | CENTERTRACK_DCN_TILE16=600 (MAIN launches below 600 workgroups on 16-pixel tiles) | =1100 (all of them)
#   3. base | -DCT_DCN_DEEP variant build (two steps of gather flight on the same two register slots)
#   4. per-launch timings of the winner (tools/dcn_slots.py)
# usage (repo root, GPU box):  bash tools/ab_dcn_r4.sh        -> gpurun_out/ab_dcn_r4/*.json(l)
# Build the variant BEFORE the call (no hipcc time on the box):  python tools/build_variant.py deep dcn_mfma.hip -DCT_DCN_DEEP
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_dcn_r4
mkdir -p $OUT
cd $R
CENTERTRACK_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_hip_experimental.py -q -x > $OUT/tests.log 2>&1
echo "experimental tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
line() {  # tag, then env assignments
    local tag=$1; shift
    for rep in 1 2; do
        env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(dict(tag='$tag', fps=j['value'], ms=j['ms_per_step'], dcn=j.get('roofline',{}))))" \
            | tee -a $OUT/ab.jsonl
    done
}
line base A=0
line tile16_600 CENTERTRACK_DCN_TILE16=600
line base A=0
line tile16_all CENTERTRACK_DCN_TILE16=1100
DEEP=$R/centertrack_amd/build/variants/libcentertrack_hip_deep.so
if [ -f $DEEP ]; then
    line base A=0
    line deep CENTERTRACK_LIB=$DEEP
    line deep_tile16 CENTERTRACK_LIB=$DEEP CENTERTRACK_DCN_TILE16=600
fi
for B in 8; do
    for v in 0 1000000000; do
        CENTERTRACK_DCN_TILE16=$v python bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(dict(tag='b$B tile16=$v', fps=j['value'], ms=j['ms_per_step'])))" | tee -a $OUT/ab.jsonl
    done
done
CENTERTRACK_DCN_TILE16=600 python tools/dcn_slots.py > $OUT/slots_tile16.txt 2>&1
python tools/dcn_slots.py > $OUT/slots_base.txt 2>&1
tail -25 $OUT/slots_tile16.txt
