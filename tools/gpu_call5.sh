#!/bin/bash
# round 3, call 5: A/B of the head-major order of the fused heads and of the DCN s_setprio variants
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3 --no-resident"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['device_ms_per_frame_batch'], 'dcn_ms', d['roofline']['total_ms'], 'conv_ms', d['roofline_conv']['total_ms'])"; }
for rep in 1 2; do
  CENTERTRACK_TUNE=heads_order=0 timeout 300 $B 2>/dev/null | show "heads_order=0"
  CENTERTRACK_TUNE=heads_order=1 timeout 300 $B 2>/dev/null | show "heads_order=1"
  timeout 300 $B 2>/dev/null | show "base        "
  CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_prio1.so timeout 300 $B 2>/dev/null | show "dcn prio1   "
  CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_prio2.so timeout 300 $B 2>/dev/null | show "dcn prio2   "
done
for s in 8; do
  timeout 300 $B --streams $s 2>/dev/null | show "b$s base     "
  CENTERTRACK_TUNE=heads_order=0 timeout 300 $B --streams $s 2>/dev/null | show "b$s heads0   "
  CENTERTRACK_LIB=$PWD/centertrack_amd/build/variants/libcentertrack_hip_prio2.so timeout 300 $B --streams $s 2>/dev/null | show "b$s prio2    "
done
