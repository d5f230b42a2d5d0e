#!/bin/bash
# round 3, call 11: 64-pixel DCN tiles at 8 / 32 streams (un-fused plans)
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --no-cpu-baseline --steps 6 --warmup 2 --no-resident"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['device_ms_per_frame_batch'], 'dcn_ms', d['roofline']['total_ms'], 'frac', d['roofline']['frac'])"; }
timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "dcn" --maxfail=3 2>&1 | tail -2
for s in 8 32; do
  timeout 300 $B --streams $s 2>/dev/null | show "b$s tile32"
  CENTERTRACK_DCN_TILE64=1 timeout 300 $B --streams $s 2>gpurun_out/r03_call11.err | show "b$s tile64"
  timeout 300 $B --streams $s 2>/dev/null | show "b$s tile32"
  CENTERTRACK_DCN_TILE64=1 timeout 300 $B --streams $s 2>>gpurun_out/r03_call11.err | show "b$s tile64"
done
tail -3 gpurun_out/r03_call11.err
