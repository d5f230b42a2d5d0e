#!/bin/bash
# round 4, second GPU call: new shapes tuned (544x960), the new / changed GPU tests, phase stamps of the conv launches,
# first bench lines at the reference's MOT size.  Outputs under gpurun_out/call2/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call2
mkdir -p $OUT
cd $R
export CENTERTRACK_TUNE_CACHE=$OUT/tune_new.json
timeout 600 python tools/tune_plans.py mot17_544x960:1 mot17_544x960:8 > $OUT/tune.log 2>&1
tail -3 $OUT/tune.log
timeout 1500 python -m pytest -q -x -m gpu tests/test_hip_rccl.py tests/test_hip_tie_policy.py tests/test_cabi.py \
    "tests/test_hip_fullsize.py::test_mot17_544x960_reference_resolution_matches_oracle" \
    "tests/test_hip_fullsize.py::test_mot17_512_T32_sequence_matches_oracle" tests/test_hip_ops.py tests/test_hip_e2e.py \
    tests/test_hip_plans.py -s > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
grep -E "tie policy|passed|failed|error" $OUT/tests.log | tail -8
timeout 300 python tools/conv_phases.py > $OUT/conv_phases_b1.txt 2>&1
cat $OUT/conv_phases_b1.txt | tail -45
for B in 1 8; do
    python bench.py --config mot17_544x960 --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 >> $OUT/bench_544.jsonl
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 > $OUT/bench_base.json
python - <<'PY'
import json
for f in ('gpurun_out/call2/bench_544.jsonl', 'gpurun_out/call2/bench_base.json'):
    for line in open(f):
        j = json.loads(line)
        print(j['config']['workload'][:60], j['value'], 'dev ms', j.get('device_ms_per_frame_batch'), 'dcn', j['roofline']['frac'], 'conv', j['roofline_conv']['frac'], j['roofline_conv'].get('algorithmic_tflops'))
PY
