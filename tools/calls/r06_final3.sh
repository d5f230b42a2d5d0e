#!/bin/bash
# round 6, final collection 3 (after the store-hazard fix): full GPU test suite, the default bench line, rocprofv3 kernel table + the
# three PMC passes of the headline command, one line + kernel table per configuration of the sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/profiles_new; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06_fin_pytest_gpu.log 2>&1; tail -3 $O/r06_fin_pytest_gpu.log
bash tools/collect_profiles.sh r06_fin > $O/r06_fin_collect.log 2>&1; tail -3 $O/r06_fin_collect.log | cut -c1-300
bash tools/sweep_configs.sh r06_fin prof > $O/r06_fin_sweep.log 2>&1; tail -14 $O/r06_fin_sweep.log
