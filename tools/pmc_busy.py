#!/usr/bin/env python
"""SQ counter summary (tools/pmc_stats.py output of the `--pmc SQ_*` pass) -> profiles/pmc_mfma_busy.json: per kernel
the share of the launch during which the matrix pipes were busy and the average number of resident waves per SIMD.

    mfma_busy          = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    waves_per_simd_avg = 4 x SQ_WAVE_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)
(GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES counts in units of 4 clocks.  Check of the formula: the
pure-MFMA calibration kernel of bench.py's `box_calibration`, `calib_mfma_kernel`, comes out at 0.99.)  Classes:
`dcn_main` = the DCN kernel named by bench.py's roofline, `conv` = every dense conv launch incl. the fused heads
(time-weighted), `backbone_3x3` = the Winograd / K-split / row-tiled 3x3 launches without the heads.
usage: pmc_busy.py <tag>_pmc_sq.txt <tag>"""
import json
import sys


def parse(path):
    rows = {}
    with open(path) as f:
        names = f.readline().split()[3:]
        idx = {}
        for want in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_INSTS_MFMA', 'SQ_WAIT_INST_ANY',
                     'SQ_BUSY_CYCLES'):
            hit = [i for i, n in enumerate(names) if want[-16:] == n or want == n]
            if hit:
                idx[want] = hit[0]
        for line in f:
            name = line[:72].strip()
            parts = line[72:].split()
            if len(parts) < 3:
                continue
            vals = [float(v) for v in parts[2:]]
            rows[name] = dict(calls=int(parts[0]), avg_us=float(parts[1]), **{k: vals[i] for k, i in idx.items()})
    return rows


def main(path, tag):
    rows = parse(path)
    out = {'source': 'profiles/%s_pmc_sq.txt (rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE, its own pass of `bench.py --steps 1 '
                     '--warmup 1 --frames-per-step 24`)' % tag,
           'formula': 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8); waves_per_simd_avg = '
                      '4 * SQ_WAVE_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)', 'kernels': {}}
    for name, r in rows.items():
        clk = r.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        if clk <= 0:
            continue
        out['kernels'][name] = {'calls': r['calls'], 'avg_us': r['avg_us'],
                                'mfma_busy': round(r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * clk), 4),
                                'waves_per_simd_avg': round(4.0 * r.get('SQ_WAVE_CYCLES', 0.0) / (1024.0 * clk), 3),
                                'wait_inst_share_of_wave_cycles': round(r.get('SQ_WAIT_INST_ANY', 0.0) / max(r.get('SQ_WAVE_CYCLES', 1.0), 1.0), 3)}

    def weighted(pred):
        sel = [(k, v) for k, v in out['kernels'].items() if pred(k)]
        t = sum(v['avg_us'] * v['calls'] for _, v in sel)
        if not t:
            return None
        return {'kernels': len(sel), 'time_share_us_per_call_weighted': round(t, 1),
                'mfma_busy': round(sum(v['mfma_busy'] * v['avg_us'] * v['calls'] for _, v in sel) / t, 4),
                'waves_per_simd_avg': round(sum(v['waves_per_simd_avg'] * v['avg_us'] * v['calls'] for _, v in sel) / t, 3)}
    dcn = [(k, v) for k, v in out['kernels'].items() if k.startswith('dcn_mfma_kernel')]
    if dcn:
        k, v = max(dcn, key=lambda kv: kv[1]['avg_us'] * kv[1]['calls'])
        out['dcn_main'] = dict(kernel=k, **v)
    is_conv = lambda k: k.startswith(('wino_conv_kernel', 'conv_mfma_kernel', 'conv_ksplit_kernel'))
    out['conv'] = weighted(is_conv)
    out['backbone_3x3'] = weighted(lambda k: is_conv(k) and 'true>(WinoArgs' not in k and not k.startswith(('conv_mfma_kernel<1,', 'conv_ksplit_kernel<1,')))
    out['calibration'] = out['kernels'].get('calib_mfma_kernel(int, float*)')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
