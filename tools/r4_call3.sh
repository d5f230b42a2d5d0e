#!/bin/bash
# round 4, third GPU call: fused projection + latency shapes.  Parity first, then the one-stream plan re-tuned from scratch
# with the new candidates (A/B against the pinned plan in the same call), then phase stamps of the re-tuned plan.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call3
mkdir -p $OUT
cd $R
timeout 900 python -m pytest -q -x -m gpu tests/test_hip_ops.py tests/test_hip_model.py > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
grep -E "passed|failed|error" $OUT/tests.log | tail -3
bench() {  # tag, env...
    local tag=$1; shift
    for rep in 1 2; do
        env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline',{}); c=j.get('roofline_conv',{}); print(json.dumps(dict(tag='$tag', fps=j['value'], dev_ms=j.get('device_ms_frame_graph'), launches=j.get('launches_per_frame'), dcn_ms=r.get('total_ms'), conv_ms=c.get('total_ms'), conv_frac=c.get('frac'))))" \
            | tee -a $OUT/ab.jsonl
    done
}
bench pinned_nofuse CENTERTRACK_FUSE_PROJ=0
export CENTERTRACK_TUNE_CACHE=$OUT/tune_b1.json
CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_DCN_KNOBS=128,4,4,1,0,0 timeout 900 python tools/tune_plans.py mot17_512:1 > $OUT/tune.log 2>&1
tail -2 $OUT/tune.log
bench retuned CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_DCN_KNOBS=128,4,4,1,0,0
bench pinned_nofuse CENTERTRACK_FUSE_PROJ=0
bench retuned CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_DCN_KNOBS=128,4,4,1,0,0
CENTERTRACK_TUNE_PINNED=0 CENTERTRACK_DCN_KNOBS=128,4,4,1,0,0 timeout 300 python tools/conv_phases.py > $OUT/conv_phases_b1.txt 2>&1
tail -42 $OUT/conv_phases_b1.txt
python - <<'PY'
import json
t = json.load(open('gpurun_out/call3/tune_b1.json'))
for k, v in sorted(t.items()):
    if k.startswith('conv'):
        print(k, v)
PY
