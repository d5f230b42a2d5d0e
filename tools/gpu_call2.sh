#!/bin/bash
# round 3, call 2: decode rewrite + native frame loop + pose variants on hardware; A/B of the loop and of the DCN small-slot knob
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_hip_model.py tests/test_hip_dropin.py -q -m gpu --maxfail=8 > gpurun_out/r03_call2_tests.log 2>&1
tail -40 gpurun_out/r03_call2_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3"
for nl in 0 1 0 1; do
  CENTERTRACK_NATIVE_LOOP=$nl timeout 300 $B 2>gpurun_out/r03_call2_loop$nl.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('native_loop=$nl', d['value'], d['resident_frames_fps'], d['device_ms_per_frame_batch'], d['host_gap_ms_per_frame_batch'])"
done
for k in 128,4,4,1,0 128,4,4,1,2 128,4,4,1,1 128,8,4,1,2 128,4,2,1,1; do
  CENTERTRACK_DCN_KNOBS=$k timeout 300 $B --no-resident 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('knobs=$k', d['value'], d['device_ms_per_frame_batch'])"
done
timeout 200 python tools/dcn_slots.py --knobs 128,4,4,1,0 > gpurun_out/r03_call2_slots_a.txt 2>&1
timeout 200 python tools/dcn_slots.py --knobs 128,4,4,1,2 > gpurun_out/r03_call2_slots_b.txt 2>&1
cat gpurun_out/r03_call2_slots_a.txt gpurun_out/r03_call2_slots_b.txt
timeout 200 python tools/dbench.py > gpurun_out/r03_call2_dbench.txt 2>&1; cat gpurun_out/r03_call2_dbench.txt
