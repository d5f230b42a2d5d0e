#!/usr/bin/env python
"""profiles/<tag>_rocprofv3_kernel_stats_*.txt (tools/rocpd_stats.py table of `rocprofv3 --kernel-trace` on bench.py)
-> profiles/dcn_profiled.json: device time of the DCN launches per frame as the PROFILER saw it, and the roofline
fraction that follows from it -- so that `roofline.frac` of the bench line can be reproduced from profiles/ alone
(bench.py attaches this file under roofline.profiled, next to its own live HIP-event figure).

    python tools/dcn_profiled.py profiles/r03_x_rocprofv3_kernel_stats_bench_steps3.txt > profiles/dcn_profiled.json

Frames = launches of stem_kernel (one per frame).  DCN kernels: dcn_mfma_kernel (MAIN launches, direct-form OFFSETS launches),
wino_offsets_kernel (round 6: the Winograd form of the OFFSETS launches) + dcn_reduce_kernel (FINISH launches).  Flops of one mot17_512 frame: 14.19 GFLOP main contraction,
+ 4.65 GFLOP offset/mask convs = 18.84 GFLOP."""
import json
import re
import sys

PEAK = 157.3
GF_MAIN, GF_ALL = 14.19, 18.84


def main(path):
    rows = {}
    with open(path) as f:
        for line in f:
            m = re.match(r'^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', line.rstrip('\n'))
            if m:
                rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    frames = [c for n, (c, _) in rows.items() if n.startswith('stem_kernel')]
    if not frames:
        raise SystemExit('no stem_kernel row in %s' % path)
    frames = frames[0]
    dcn = {n: v for n, v in rows.items() if n.startswith(('dcn_mfma_kernel', 'dcn_reduce_kernel', 'wino_offsets_kernel'))}
    total_us = sum(t for _, t in dcn.values())
    per_frame = total_us / frames
    main_us = sum(t for n, (_, t) in dcn.items() if n.startswith('dcn_mfma_kernel')) / frames
    out = {
        'source': path, 'frames': frames,
        'dcn_us_per_frame': round(per_frame, 1),
        'dcn_mfma_kernel_us_per_frame': round(main_us, 1),
        'launches_per_frame': round(sum(c for c, _ in dcn.values()) / frames, 2),
        'kernels': {n: {'calls_per_frame': round(c / frames, 2), 'avg_us': round(t / c, 2)} for n, (c, t) in sorted(dcn.items())},
        'tflops': round(GF_ALL / per_frame * 1e3, 2), 'frac': round(GF_ALL / per_frame * 1e3 / PEAK, 4),
        'tflops_main_formula': round(GF_MAIN / per_frame * 1e3, 2), 'frac_main': round(GF_MAIN / per_frame * 1e3 / PEAK, 4),
        'note': 'kernel durations under rocprofv3 --kernel-trace (slightly inflated by the tracer); flops per frame: '
                '%.2f G main contraction + offset/mask convs = %.2f G (SURVEY.md 8d)' % (GF_MAIN, GF_ALL),
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
