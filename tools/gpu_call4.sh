#!/bin/bash
# round 3, call 4: host flag + split stem A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_e2e.py tests/test_hip_model.py -q -m gpu --maxfail=5 > gpurun_out/r03_call4_tests.log 2>&1
tail -4 gpurun_out/r03_call4_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3"
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 4" "1 4"; do
  set -- $cfg
  CENTERTRACK_HOST_FLAG=$1 CENTERTRACK_SPLIT_STEM_MAX=$2 timeout 300 $B 2>gpurun_out/r03_call4.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flag=$1 split=$2', d['value'], d['resident_frames_fps'], d['device_ms_per_frame_batch'], d.get('device_ms_frame_graph'), d.get('device_ms_prestage'), d['host_gap_ms_per_frame_batch'])"
done
done
