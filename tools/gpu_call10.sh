#!/bin/bash
# round 3, call 10: helper threads of the native loop (test + A/B at 8 / 16 / 32 streams)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_e2e.py -q -m gpu --maxfail=5 > gpurun_out/r03_call10_tests.log 2>&1
tail -4 gpurun_out/r03_call10_tests.log
B="python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --no-resident"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['device_ms_per_frame_batch'], 'gap', d['host_gap_ms_per_frame_batch'])"; }
for s in 8 16 32; do
  for th in 1 0; do
    if [ $th = 1 ]; then export CENTERTRACK_HOST_THREADS=1; else unset CENTERTRACK_HOST_THREADS; fi
    timeout 300 $B --streams $s 2>/dev/null | show "b$s threads=${th}(0=default)"
    timeout 300 $B --streams $s 2>/dev/null | show "b$s threads=${th}(0=default)"
  done
done
unset CENTERTRACK_HOST_THREADS
timeout 300 $B --config nusc_800x448 --streams 32 2>/dev/null | show "nusc b32 default"
CENTERTRACK_HOST_THREADS=1 timeout 300 $B --config nusc_800x448 --streams 32 2>/dev/null | show "nusc b32 threads=1"
