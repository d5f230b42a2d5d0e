#!/bin/bash
# round 3, call 7: A/B of the per-slot un-fuse policy of the DCN schedule (knobs 5 / 6)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3 --no-resident"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['device_ms_per_frame_batch'], 'dcn_ms', d['roofline']['total_ms'], 'launches', d['launches_per_frame'])"; }
for rep in 1 2; do
for k in 128,4,4,1,0,0 128,4,4,1,2,1 128,4,4,1,2,2 128,4,4,1,0,2 128,8,4,1,2,2 64,4,4,1,2,2 128,8,4,1,0,2; do
  CENTERTRACK_DCN_KNOBS=$k timeout 300 $B 2>/dev/null | show "knobs=$k"
done
done
timeout 200 python tools/dcn_slots.py --knobs 128,4,4,1,2,2 > gpurun_out/r03_call7_slots_a.txt 2>&1
timeout 200 python tools/dcn_slots.py --knobs 128,4,4,1,0,2 > gpurun_out/r03_call7_slots_b.txt 2>&1
cat gpurun_out/r03_call7_slots_a.txt gpurun_out/r03_call7_slots_b.txt | grep -v amdgpu.ids
