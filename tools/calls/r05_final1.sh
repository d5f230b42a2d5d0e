#!/bin/bash
# final evidence, part 1 (fast node only: host kernel 6.18.51): bench line + rocprofv3 kernel stats + PMC passes + config sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
K=$(python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print(box_calib.node().get('kernel'))")
echo "host kernel $K"
if [ "$K" == "6.18.50-ant.1" ] && [ "${1:-}" != "any" ]; then echo "slow node, not collecting"; exit 0; fi
timeout 900 bash tools/collect_profiles.sh r05_b 2>&1 | tail -3
timeout 1200 bash tools/sweep_configs.sh r05_b prof 2>&1 | tail -16
