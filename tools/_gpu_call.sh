cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fullsize.py -q -m gpu -x 2>&1 | tail -4
