#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call8
mkdir -p $OUT
cd $R
timeout 900 python -m pytest -q -x -m gpu tests/test_hip_ops.py tests/test_hip_model.py > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
V=$R/centertrack_amd/build/variants/libcentertrack_hip_base.so
ab() {
    local tag=$1; shift
    for rep in 1 2; do
        env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | \
            python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline',{}); c=j.get('roofline_conv',{}); print(json.dumps(dict(tag='$tag', fps=j['value'], dev_ms=j.get('device_ms_frame_graph'), dcn_ms=r.get('total_ms'), conv_ms=c.get('total_ms'))))" \
            | tee -a $OUT/ab_prefetch.jsonl
    done
}
ab base CENTERTRACK_LIB=$V
ab prefetch A=0
ab base CENTERTRACK_LIB=$V
ab prefetch A=0
python tools/conv_phases.py > $OUT/conv_phases_b1.txt 2>&1
python tools/dcn_phases.py > $OUT/dcn_phases_b1.txt 2>&1
grep -E "level3.tree2|level5|level2.t2|heads" $OUT/conv_phases_b1.txt | cut -c1-140
grep -E "layer 0" $OUT/dcn_phases_b1.txt | tail -3 | cut -c1-260
for B in 8 32; do
  for lib in $V ""; do
    CENTERTRACK_LIB=$lib python bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('b$B', '${lib:+base}', j['value'], j.get('device_ms_frame_graph'))"
  done
done
