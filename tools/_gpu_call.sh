cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CENTERTRACK_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r02d.json
(timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "heads") 2>&1 | tail -4
(timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_e2e.py tests/test_hip_dropin.py -m gpu -x -q) 2>&1 | tail -4
for fh in 1 0 1 0; do
  CENTERTRACK_FUSE_HEADS=$fh python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fuse_heads=$fh B=1', j['value'], j['resident_frames_fps'], j['device_ms_per_frame_batch'], j['launches_per_frame'], j['roofline_conv']['total_ms'])"
done
for fh in 1 0; do
  CENTERTRACK_FUSE_HEADS=$fh python bench.py --no-cpu-baseline --streams 8 --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fuse_heads=$fh B=8', j['value'], j['resident_frames_fps'], j['device_ms_per_frame_batch'], j['launches_per_frame'], j['roofline_conv']['total_ms'])"
done
