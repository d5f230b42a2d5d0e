cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CENTERTRACK_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r02d.json
(timeout 900 python -m pytest tests/test_hip_e2e.py -m gpu -x -q -k "prefetch or path or pose") 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=1', j['value'], j['resident_frames_fps'], j['device_ms_per_frame_batch'])"
done
python bench.py --no-cpu-baseline --streams 8 --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=8', j['value'], j['resident_frames_fps'], j['device_ms_per_frame_batch'])"
python bench.py --no-cpu-baseline --streams 32 --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B=32', j['value'], j['resident_frames_fps'], j['device_ms_per_frame_batch'])"
