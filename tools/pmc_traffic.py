#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_stats.py output) -> profiles/pmc_traffic.json: HBM bytes per layer
of the DCNv2 kernels, corrected as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE taken as is
(checked against kernels whose output size is known: heads conv 81920 KiB = its 83.9 MB output, stem 16384 KiB)."""
import json
import sys


def parse(path, counter):
    rows = {}
    with open(path) as f:
        header = f.readline().split()
        names = header[3:]
        col = [i for i, n in enumerate(names) if counter[-16:] in n][0]
        for line in f:
            name = line[:72].strip()
            parts = line[72:].split()
            if len(parts) < 3:
                continue
            rows[name] = (int(parts[0]), float(parts[1]), float(parts[2 + col]))
    return rows


def main(fetch_path, write_path, tag):
    fr, wr = parse(fetch_path, 'FETCH_SIZE'), parse(write_path, 'WRITE_SIZE')
    out = {'source': 'profiles/%s_pmc_fetch_size.txt + profiles/%s_pmc_write_size.txt (rocprofv3 --pmc, separate passes of '
                     '`bench.py --steps 1 --warmup 1 --frames-per-step 24`)' % (tag, tag),
           'correction': 'KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read under-count, MI355X_MICROARCH.md); WRITE_SIZE as is',
           'kernels': {}}
    for name in fr:
        if name in wr and ('dcn_' in name or 'conv_' in name or 'stem' in name or 'wino_offsets' in name):
            calls, us, fkib = fr[name]
            wkib = wr[name][2]
            out['kernels'][name] = {'calls': calls, 'avg_us': us, 'fetch_bytes': 2 * 1024 * fkib, 'write_bytes': 1024 * wkib,
                                    'hbm_bytes_per_launch': 2 * 1024 * fkib + 1024 * wkib}
    # every DCN kernel of a frame (grouped gather + contraction launches, the finishing reduce / IDAUp launches), per
    # LAYER: 16 DCNv2 layers per frame; frames = launches of the stem kernel (one per frame)
    dcn = [v for k, v in out['kernels'].items() if 'dcn_' in k or 'wino_offsets' in k]
    frames = [v['calls'] for k, v in out['kernels'].items() if 'stem' in k]
    if dcn and frames:
        total = sum(v['hbm_bytes_per_launch'] * v['calls'] for v in dcn)
        out['dcn_mfma_kernel'] = {'frames': frames[0], 'launches_per_frame': sum(v['calls'] for v in dcn) / frames[0],
                                  'hbm_bytes_per_frame': total / frames[0], 'hbm_bytes_per_layer': total / frames[0] / 16}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
