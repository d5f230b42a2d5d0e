#!/bin/bash
# Looks at the lease's box state first (instruction-fetch probe of tools/box_calib.py + a short headline run); only a SLOW lease
# (DESIGN.md section 4 "Box states") goes on to the scripted instruction-cache counter passes (tools/pmc_ifetch.sh) and a kernel table.
# A fast lease costs about a minute of box time and leaves one line in gpurun_out/r06_slowhunt/leases.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_slowhunt; mkdir -p $O
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-roofline --no-resident > $O/probe_bench.json 2> $O/probe_bench.err
LINE=$(python - <<PY
import json
j=json.loads([l for l in open('$O/probe_bench.json') if l.startswith('{')][-1])
st=j.get('box_calibration',{}).get('state',{})
dev=j.get('device_ms_per_frame_batch',0)
slow = st.get('instruction_fetch')=='slow' or dev > 1.0
print('slow' if slow else 'fast', dev, st.get('ifetch_64KB_code_256wg_us'), j.get('value'))
PY
)
echo "$(date -u +%H:%M:%S) $LINE" | tee -a $O/leases.txt
if [ "$(echo $LINE | cut -d' ' -f1)" = slow ]; then
  bash tools/pmc_ifetch.sh r06slow
  mkdir -p $O/ifetch; cp gpurun_out/r05_ifetch/r06slow_* $O/ifetch/ 2>/dev/null
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_slow
  rocprofv3 --kernel-trace --stats -d /tmp/prof_slow -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_slow/*/*.db | head -1) 40 > $O/slow_kernel_stats_bench_steps3.txt
  cd $R; python tools/kbench.py --batch 1 --no-dcn --layers "3x3 " --variant auto > $O/slow_kbench_b1.txt 2>&1
fi
