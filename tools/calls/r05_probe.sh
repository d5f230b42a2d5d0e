#!/bin/bash
# one cheap call: the box probes + the headline bench (dense) -> does a probe separate the classes?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; TAG=${1:-x}; mkdir -p gpurun_out/r05_probe; O=$R/gpurun_out/r05_probe
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench_$TAG.json 2>$O/bench_$TAG.err
python - <<P
import json
j = json.loads([l for l in open('$O/bench_$TAG.json') if l.startswith('{')][-1])
b = j['box_calibration']
print('$TAG', j['value'], 'fps dev', j['device_ms_per_frame_batch'], 'dcn', j['roofline']['total_ms'], 'conv', j['roofline_conv']['total_ms'], b['sysfs'].get('dir'))
print({k: v for k, v in b["launch_us"].items()})
P
