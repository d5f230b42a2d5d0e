#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_b; O=gpurun_out/r05_b
python tools/box_calib.py > $O/box.json 2>/dev/null; cat $O/box.json
timeout 900 python -m pytest tests/test_hip_e2e.py tests/test_hip_model.py tests/test_hip_rccl.py -x -q -k "modes or full_size or rccl" 2>&1 | tail -25 | tee $O/tests.log
