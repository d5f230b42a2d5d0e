#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_g; O=$R/gpurun_out/r05_g
python -c "
import sys; sys.path.insert(0,'.')
from tools import box_calib; print(box_calib.node().get('kernel'))"
for rep in 1 2; do
echo "== base"; python tools/dcn_slots.py 2>/dev/null | grep -v "^knobs" | awk '{print $(NF-7), $(NF-6), $(NF-3), $(NF-2)}' | tr '\n' ';'; echo
echo "== offpd8"; CENTERTRACK_LIB=$R/centertrack_amd/build/variants/libcentertrack_hip_offpd8.so python tools/dcn_slots.py 2>/dev/null | grep -v "^knobs" | awk '{print $(NF-7), $(NF-6), $(NF-3), $(NF-2)}' | tr '\n' ';'; echo
done
