#!/bin/bash
# First GPU call of round 4: the experiments prepared (blind) at the end of round 3, as A/B pairs inside ONE call.
#   0. parity of the 16-pixel DCN shape (algo 41664) -- tests/test_hip_experimental.py; its lines below count only if green
#   1. base | CENTERTRACK_DCN_TILE16=600 (MAIN launches below 600 workgroups on 16-pixel tiles) | =1100 (all of them)
#      | CENTERTRACK_DCN_M32=1 (MAIN launches on 32 x 32 x 2 MFMA tiles)
#   2. variant builds (bit-identical kernels, other resource use / staging order):
#        w4      -DCT_DCN_WAVES=4     DCN kernels capped at 128 VGPRs (148 -> 128, no spills): 4 workgroups per CU
#        deep    -DCT_DCN_DEEP        two steps of gather flight on the same two register slots
#        stemw4  -DCT_STEM_WAVES=4    stem capped at 128 VGPRs (140 -> 124): its 1024 workgroups in one round instead of 1.33
#        stemco  -DCT_STEM_COALESCED  stem weights staged along k (coalesced) into a 17-float LDS pitch
#        all     w4 + stemw4 + stemco
#   3. per-launch timings of base / tile16 / w4 (tools/dcn_slots.py)
# BEFORE the call, here (no hipcc time on the box):   bash tools/ab_dcn_r4.sh build
# on the box:                                         bash tools/ab_dcn_r4.sh        -> gpurun_out/ab_dcn_r4/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/centertrack_amd/build/variants
if [ "${1:-}" = build ]; then
    cd $R
    python tools/build_variant.py w4 dcn_mfma.hip -DCT_DCN_WAVES=4
    python tools/build_variant.py deep dcn_mfma.hip -DCT_DCN_DEEP
    python tools/build_variant.py stemw4 stem.hip -DCT_STEM_WAVES=4
    python tools/build_variant.py stemco stem.hip -DCT_STEM_COALESCED
    python tools/build_variant.py all dcn_mfma.hip -DCT_DCN_WAVES=4 + stem.hip -DCT_STEM_WAVES=4 -DCT_STEM_COALESCED
    exit 0
fi
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
            python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline',{}); print(json.dumps(dict(tag='$tag', fps=j['value'], ms=j['ms_per_step'], dcn_frac=r.get('frac'), dcn_layer_us=r.get('avg_launch_us'), dev_ms=j.get('device_ms_frame_graph'))))" \
            | tee -a $OUT/ab.jsonl
    done
}
line base A=0
line tile16_600 CENTERTRACK_DCN_TILE16=600
line tile16_all CENTERTRACK_DCN_TILE16=1100
line m32 CENTERTRACK_DCN_M32=1                              # MAIN launches on v_mfma_f32_32x32x2_f32 tiles (algo 53264)
for v in w4 deep stemw4 stemco all; do
    if [ -f $V/libcentertrack_hip_$v.so ]; then
        line base A=0
        line $v CENTERTRACK_LIB=$V/libcentertrack_hip_$v.so
        # the variant kernels are bit-identical by construction: the ops suite says whether they are
        CENTERTRACK_LIB=$V/libcentertrack_hip_$v.so timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "dcn or stem" > $OUT/tests_$v.log 2>&1
        echo "$v ops tests rc=$?" | tee -a $OUT/tests.log
    fi
done
# the un-fused schedule (every offset conv in its slot's K-split OFFSETS launch: round 3's experiment 2, MAIN launches at
# 75-90 TFLOP/s but 8.5-9 us per OFFSETS launch) with those launches on one-row tiles (knob dcn_offs16), alone and with
# the 16-pixel MAIN tiles
line base A=0
line unfused CENTERTRACK_DCN_KNOBS=0,4,4,1,0,0
line unfused_offs16 CENTERTRACK_DCN_KNOBS=0,4,4,1,0,0 CENTERTRACK_TUNE=dcn_offs16=1
line unfused_offs16_tile16 CENTERTRACK_DCN_KNOBS=0,4,4,1,0,0 CENTERTRACK_TUNE=dcn_offs16=1 CENTERTRACK_DCN_TILE16=1100
line offs16 CENTERTRACK_TUNE=dcn_offs16=1
if [ -f $V/libcentertrack_hip_w4.so ]; then line w4_tile16 CENTERTRACK_LIB=$V/libcentertrack_hip_w4.so CENTERTRACK_DCN_TILE16=600; fi
for B in 8; do
    for v in 0 1000000000; do
        CENTERTRACK_DCN_TILE16=$v python bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(dict(tag='b$B tile16=$v', fps=j['value'], ms=j['ms_per_step'])))" | tee -a $OUT/ab.jsonl
    done
    CENTERTRACK_DCN_M32=1 python bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
        python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(dict(tag='b$B m32', fps=j['value'], ms=j['ms_per_step'])))" | tee -a $OUT/ab.jsonl
    if [ -f $V/libcentertrack_hip_w4.so ]; then
        CENTERTRACK_LIB=$V/libcentertrack_hip_w4.so python bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(dict(tag='b$B w4', fps=j['value'], ms=j['ms_per_step'])))" | tee -a $OUT/ab.jsonl
    fi
done
python tools/dcn_slots.py > $OUT/slots_base.txt 2>&1
CENTERTRACK_DCN_TILE16=600 python tools/dcn_slots.py > $OUT/slots_tile16.txt 2>&1
[ -f $V/libcentertrack_hip_w4.so ] && CENTERTRACK_LIB=$V/libcentertrack_hip_w4.so python tools/dcn_slots.py > $OUT/slots_w4.txt 2>&1
tail -25 $OUT/slots_tile16.txt
