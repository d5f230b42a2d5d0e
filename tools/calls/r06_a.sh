#!/bin/bash
# round 6, call A: (1) the instruction-cache counter passes of the headline frame, (2) per-launch timing
# of the DCN schedule at 4 streams, (3) the DCN tile shapes layer by layer at 4 and 8 streams (tools/kbench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_a; mkdir -p $O
bash tools/pmc_ifetch.sh r06a > $O/ifetch.log 2>&1
cp -r gpurun_out/r05_ifetch $O/ 2>/dev/null
python tools/dcn_slots.py --batch 4 --size 512 > $O/dcn_slots_b4.txt 2>&1
python tools/dcn_slots.py --batch 1 --size 512 > $O/dcn_slots_b1.txt 2>&1
python tools/kbench.py --batch 4 --no-conv > $O/kbench_dcn_b4.txt 2>&1
python tools/kbench.py --batch 8 --no-conv > $O/kbench_dcn_b8.txt 2>&1
python tools/kbench.py --batch 4 --no-dcn --layers '3x3 ' > $O/kbench_conv_b4.txt 2>&1
tail -30 $O/kbench_dcn_b4.txt
