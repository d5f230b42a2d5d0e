#!/bin/bash
# tie / threshold-flip report of the final kernels (after the store-hazard fix): 1760 full-size frames, nothing re-seeded
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_au; mkdir -p $O
timeout 2400 python tools/tie_report.py --out $O/r06_tie_report.json --dump $O/tie_dump.pkl > $O/tie.log 2>&1; tail -25 $O/tie.log | cut -c1-220
ls -la $O/tie_dump.pkl; [ $(stat -c %s $O/tie_dump.pkl) -gt 50000000 ] && rm -f $O/tie_dump.pkl
