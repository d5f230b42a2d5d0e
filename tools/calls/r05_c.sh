#!/bin/bash
# round 5, call C: does the DPM governor explain the box classes?  bench at perf level auto vs high (restored afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_c; O=gpurun_out/r05_c
D=$(python - <<'P'
import torch, os
pr = torch.cuda.get_device_properties(0)
print('/sys/bus/pci/devices/%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
P
)
echo "device dir $D"; cat $D/power_dpm_force_performance_level; ls -la $D/power_dpm_force_performance_level
trap 'echo auto > $D/power_dpm_force_performance_level 2>/dev/null' EXIT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes"
show() { python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', j['value'], 'fps resident', j['resident_frames_fps'], 'dev', j['device_ms_per_frame_batch'], 'dcn', j['roofline']['total_ms'], 'conv', j['roofline_conv']['total_ms'], json.dumps(j['box_calibration'].get('clocks_under_load')), 'd2d', j['box_calibration']['d2d_1GiB_GBps'])"; }
$B 2>/dev/null | show auto1
for lvl in high auto; do
  if echo $lvl > $D/power_dpm_force_performance_level 2>$O/set_$lvl.err; then
    echo "set $lvl ok: $(cat $D/power_dpm_force_performance_level) sclk: $(cat $D/pp_dpm_sclk | tr '\n' ' ')"
    $B 2>/dev/null | show $lvl
  else
    echo "cannot set $lvl: $(cat $O/set_$lvl.err)"
  fi
done
python tools/box_calib.py 2>/dev/null | tail -1 > $O/box.json; cat $O/box.json
