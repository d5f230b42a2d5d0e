#!/bin/bash
# round 3, call 9: decode -> host rows + flag (tests, A/B), offset-conv prefetch-depth variants
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py -q -m gpu --maxfail=6 -k "decode or native or stream or gather or pose or detector" > gpurun_out/r03_call9_tests.log 2>&1
tail -5 gpurun_out/r03_call9_tests.log
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3 --no-resident"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['device_ms_per_frame_batch'], 'gap', d['host_gap_ms_per_frame_batch'], 'dcn_ms', d['roofline']['total_ms'], 'launches', d['launches_per_frame'])"; }
for rep in 1 2; do
  CENTERTRACK_HOST_ROWS=0 timeout 300 $B 2>/dev/null | show "host_rows=0 "
  CENTERTRACK_HOST_ROWS=1 timeout 300 $B 2>/dev/null | show "host_rows=1 "
  CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_offpd5.so timeout 300 $B 2>/dev/null | show "offpd5      "
  CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_offpd8.so timeout 300 $B 2>/dev/null | show "offpd8      "
done
CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_offpd8.so CENTERTRACK_DCN_KNOBS=128,4,4,1,2,2 timeout 300 $B 2>/dev/null | show "offpd8 unfuse"
CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_offpd8.so timeout 200 python tools/dcn_slots.py > gpurun_out/r03_call9_slots_pd8.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03_call9_slots_pd8.txt
timeout 300 $B --streams 8 2>/dev/null | show "b8 base     "
CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_offpd8.so timeout 300 $B --streams 8 2>/dev/null | show "b8 offpd8   "
