#!/usr/bin/env python
"""Append the probe calls of ``tools/calls/r05_probe.sh`` (gpurun_out/r05_probe/bench_<tag>.json) to
``profiles/r05_a_box_probes.jsonl``: one row per call = the headline numbers of that lease + the box probes beside them.

    python tools/export_box_probes.py [--dir gpurun_out/r05_probe] [--out profiles/r05_a_box_probes.jsonl]

Rows already in the file (by ``call``) are left alone; a summary of the two states is printed.
"""
import argparse
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row_of(tag, j):
    b = j['box_calibration']
    sysfs = b.get('sysfs') or {}
    node = b.get('node') or {}
    temps = {k: v for k, v in sysfs.items() if k.startswith('temp_')} or None
    cu = (b.get('cu_map') or {}).get('512wg_2_per_cu_20us')
    return {
        'call': tag, 'fps': j['value'], 'device_ms': j['device_ms_per_frame_batch'], 'dcn_ms': j['roofline']['total_ms'],
        'conv_ms': j['roofline_conv']['total_ms'], 'pci': os.path.basename(sysfs.get('dir', '')) or None,
        'host_kernel': node.get('kernel'), 'gpu_numa_node': node.get('gpu_numa_node'), 'temps_mC': temps,
        'mfma_f32_tflops': b.get('mfma_f32_tflops'), 'd2d_1GiB_GBps': b.get('d2d_1GiB_GBps'), 'chase_ns': b.get('chase_ns'),
        'stream_GBps': b.get('stream_GBps'), 'launch_us': b.get('launch_us'),
        'sclk_under_load': (b.get('clocks_under_load') or {}).get('sclk_mhz') or b.get('sclk_under_load'),
        'cu_map_512wg': cu, 'xcd_stream': b.get('xcd_stream'),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default=os.path.join(ROOT, 'gpurun_out', 'r05_probe'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_a_box_probes.jsonl'))
    a = ap.parse_args()
    rows = [json.loads(l) for l in open(a.out)] if os.path.exists(a.out) else []
    have = {r['call'] for r in rows}
    files = sorted(glob.glob(os.path.join(a.dir, 'bench_t*.json')), key=lambda p: int(re.search(r't(\d+)', p).group(1)))
    for p in files:
        tag = re.search(r'bench_(t\d+)\.json', p).group(1)
        if tag in have:
            continue
        lines = [l for l in open(p) if l.startswith('{')]
        if lines:
            rows.append(row_of(tag, json.loads(lines[-1])))
    with open(a.out, 'w') as f:
        for r in rows:
            f.write(json.dumps(r) + '\n')
    slow = [r for r in rows if r['fps'] < 960]
    fast = [r for r in rows if r['fps'] >= 960]
    for name, rs in (('fast', fast), ('slow', slow)):
        ifetch = [r['launch_us']['ifetch_64KB_code_256wg'] for r in rs if 'ifetch_64KB_code_256wg' in (r.get('launch_us') or {})]
        print('%s: %d calls, fps %.0f-%.0f, device %.3f-%.3f ms, ifetch_256wg %s' % (
            name, len(rs), min(r['fps'] for r in rs), max(r['fps'] for r in rs), min(r['device_ms'] for r in rs),
            max(r['device_ms'] for r in rs), sorted(ifetch)))


if __name__ == '__main__':
    main()
