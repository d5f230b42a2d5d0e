#!/bin/bash
# round 3, call 3: decode histogram select, split stem + pre-stage, native loop again; A/B of the split stem
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_hip_model.py tests/test_hip_dropin.py -q -m gpu --maxfail=8 > gpurun_out/r03_call3_tests.log 2>&1
tail -15 gpurun_out/r03_call3_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3"
for sp in 0 4 0 4; do
  CENTERTRACK_SPLIT_STEM_MAX=$sp timeout 300 $B 2>gpurun_out/r03_call3_split$sp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split_stem_max=$sp', d['value'], d['resident_frames_fps'], d['device_ms_per_frame_batch'], d.get('device_ms_frame_graph'), d.get('device_ms_prestage'), d['host_gap_ms_per_frame_batch'])"
done
CENTERTRACK_SPLIT_STEM_MAX=4 timeout 300 $B --streams 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b4 split', d['value'], d['device_ms_per_frame_batch'])"
CENTERTRACK_SPLIT_STEM_MAX=0 timeout 300 $B --streams 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b4 nosplit', d['value'], d['device_ms_per_frame_batch'])"
timeout 200 python tools/dbench.py > gpurun_out/r03_call3_dbench.txt 2>&1; cat gpurun_out/r03_call3_dbench.txt
