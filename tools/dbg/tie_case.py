"""One stream of tools/tie_report.py replayed frame by frame, both sides printed around one id divergence (debugging aid).
    python tools/dbg/tie_case.py mot17_512 0 17 0 4        # config, plan index, run, stream, frames to run, frames of the stream"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch


def main(name, plan, run, s, T, TT):
    import scenarios as S
    import tie_report as TR
    from _parity import calibrated_state_dict, scrolled_stream
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP
    from oracle import detector as odet
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    H, W = cfg['H'], cfg['W']
    sd = calibrated_state_dict(name, heads)
    kw = dict(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=cfg['flip'])
    opt = default_opt(heads, **kw)
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=1)
    oopt = odet.default_opt(input_h=H, input_w=W, num_classes=heads['hm'], **kw)
    orc = odet.Detector(oopt, sd, heads)
    meta = make_meta(H, W, 2 * H, 2 * W)
    frames = scrolled_stream(H, W, TT, TR.stream_seed(plan, run, s))      # (the base image depends on the stream's full length)
    prev = None
    for t in range(T):
        res = det.step(frames[t], [dict(meta)])
        got = det.results_as_dicts(res[0], 0, meta)
        want = orc.run(frames[t], dict(meta))
        gd, od = det.last_dets, orc.last_dets
        def key(r):
            return tuple(np.round(np.asarray(r['bbox'], np.float64) / 4).astype(int).tolist())
        gm = {key(r): r for r in got}
        print('frame %d: %d hip results, %d oracle results' % (t, len(got), len(want)))
        for r in want:
            g = gm.get(key(r))
            flag = '' if g is not None and int(g['tracking_id']) == int(r['tracking_id']) else '   <-- differs'
            if flag:
                print('  oracle id %3d score %.6f bbox %s ct %s tracking %s | hip %s%s' % (
                    r['tracking_id'], r['score'], np.round(np.asarray(r['bbox']), 3).tolist(), np.round(np.asarray(r['ct']), 4).tolist(),
                    np.round(np.asarray(r['tracking']), 4).tolist(),
                    None if g is None else 'id %3d score %.6f bbox %s ct %s tracking %s' % (
                        g['tracking_id'], g['score'], np.round(np.asarray(g['bbox']), 3).tolist(), np.round(np.asarray(g['ct']), 4).tolist(),
                        np.round(np.asarray(g['tracking']), 4).tolist()), flag))
        if prev is not None and t == T - 1:
            print('  previous frame tracks (oracle): ')
            for r in prev[1]:
                print('     id %3d score %.6f ct %s bbox %s' % (r['tracking_id'], r['score'], np.round(np.asarray(r['ct']), 4).tolist(),
                                                                np.round(np.asarray(r['bbox']), 3).tolist()))
            print('  previous frame tracks (hip): ')
            for r in prev[0]:
                print('     id %3d score %.6f ct %s bbox %s' % (r['tracking_id'], r['score'], np.round(np.asarray(r['ct']), 4).tolist(),
                                                                np.round(np.asarray(r['bbox']), 3).tolist()))
        prev = (got, want)


if __name__ == '__main__':
    a = sys.argv
    main(a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]) if len(a) > 6 else int(a[5]))
