"""Run-to-run identity of the frame path: the same seeded streams through FRESH detectors, several times in one
process, every frame's head maps and packed rows compared BITWISE with the first pass.

No kernel of the path uses a floating-point atomic and every launch shape is pinned (tune_table.json), so two passes over the
same frames must agree in every bit.  A difference is a race or a read of memory nobody wrote; to make the second kind
visible the allocator's free blocks are filled with NaNs between the passes (``--poison``, default on): a detector that is
built on recycled memory and reads a word it never wrote returns NaNs instead of stale-but-plausible numbers.

    python tools/determinism.py --config coco_512 --streams 4 --frames 8 --passes 4

prints one JSON line per (config, pass): first differing frame, the differing heads with max |diff| and the count of
differing rows; exit status 1 if any pass differs.  tests/test_hip_determinism.py runs the same comparison in the GPU suite.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))

import numpy as np
import torch


def poison_free_memory(device, gib):
    """fill ``gib`` GiB of the caching allocator's pool with NaNs and hand the blocks back: the next tensors of the process
    are carved out of them"""
    if gib <= 0:
        return
    blocks = []
    for sz in (512 << 20, 64 << 20, 8 << 20, 1 << 20):
        n = max(1, int(gib * (1 << 30) / 4 / sz))
        for _ in range(n):
            try:
                blocks.append(torch.full((sz // 4,), float('nan'), device=device))
            except RuntimeError:
                break
    torch.cuda.synchronize()
    del blocks


def one_pass(name, streams, T, seed0, keep_heads, sparse_heads=False):
    """T frames through a fresh model + StreamDetector; returns per-frame dicts {head: ndarray} / rows / result ids"""
    import scenarios as S
    from _parity import calibrated_state_dict, scrolled_stream
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    H, W = cfg['H'], cfg['W']
    opt = default_opt(heads, track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=cfg['flip'], sparse_heads=sparse_heads)
    model = DLASegHIP(heads)
    model.load_state_dict(calibrated_state_dict(name, heads))
    det = StreamDetector(opt, model=model, num_streams=streams)
    meta = make_meta(H, W, 2 * H, 2 * W)
    frames = [scrolled_stream(H, W, T, seed0 + 100 * s) for s in range(streams)]
    out = []
    for t in range(T):
        res = det.step(torch.cat([frames[s][t] for s in range(streams)], 0), [dict(meta) for _ in range(streams)])
        rec = {'rows': {k: np.array(v) for k, v in det.last_dets.items()},
               'ids': [[int(r['tracking_id']) for r in det.results_as_dicts(res[s], s, meta)] for s in range(streams)]}
        if keep_heads:
            torch.cuda.synchronize()
            rec['heads'] = {k: v.detach().cpu().numpy().copy() for k, v in det._ctx['merged'].items()}
        out.append(rec)
    del det, model
    return out


def diff_passes(a, b):
    """first frame at which pass ``b`` differs from pass ``a`` + what differs there (None: identical)"""
    for t, (x, y) in enumerate(zip(a, b)):
        bad = {}
        for grp in ('heads', 'rows'):
            for k in x.get(grp, {}):
                u, v = x[grp][k], y[grp][k]
                if u.tobytes() != v.tobytes():
                    d = np.abs(u.astype(np.float64) - v.astype(np.float64))
                    nan = int(np.isnan(d).sum())
                    bad['%s.%s' % (grp, k)] = {'max': float(np.nanmax(d)) if nan < d.size else None, 'n': int((d > 0).sum()),
                                               'nan': nan,
                                               'images': sorted(set(np.argwhere(~(d == 0))[:, 0].tolist()))[:8]}
        if x['ids'] != y['ids']:
            bad['ids'] = {'streams': [s for s in range(len(x['ids'])) if x['ids'][s] != y['ids'][s]]}
        if bad:
            return t, bad
    return None


def _walk(obj, found, seen):
    """every torch tensor reachable from a launch's ``keep`` (views by their buffers)"""
    from centertrack_amd.ops import View
    if obj is None or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, View):
        obj = obj.buf
    if torch.is_tensor(obj):
        if obj.is_cuda and obj.dtype == torch.float32:
            found.append(obj)
        return
    if isinstance(obj, dict):
        return              # (packed weights: constant)
    if isinstance(obj, (list, tuple)):
        for o in obj:
            _walk(o, found, seen)


def plan_buffers(plan):
    """[(launch index, launch name, k, tensor)] in launch order, every buffer once (by address), named by the first launch
    that refers to it"""
    out, ptrs, seen = [], set(), set()
    for i, l in enumerate(plan['launches']):
        found = []
        _walk(l.keep, found, seen)
        if l.fn == 'stem':
            _walk(l.args, found, seen)
        for k, t in enumerate(found):
            base = t.untyped_storage().data_ptr()
            if base in ptrs:
                continue
            ptrs.add(base)
            out.append((i, l.name, k, t))
    for k, t in plan['outputs'].items():
        base = t.untyped_storage().data_ptr()
        if base not in ptrs:
            ptrs.add(base)
            out.append((len(plan['launches']), 'outputs.' + k, 0, t))
    return out


def where(d):
    """bounding box of the non-zero entries of a difference array"""
    idx = np.argwhere(d != 0)
    return [[int(idx[:, a].min()), int(idx[:, a].max())] for a in range(idx.shape[1])]


def model_mode(name, streams, runs, use_graph, inputs=2):
    """the forward alone: ``inputs`` seeded input sets visited round-robin, every buffer of the plan compared with its copy
    from the first visit of the same input set.  Reports, per differing run, the buffers that differ in launch order -- the
    first one names the launch whose result is not a function of its inputs."""
    import scenarios as S
    from _parity import calibrated_state_dict
    from centertrack_amd.model import DLASegHIP
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    H, W = cfg['H'], cfg['W']
    N = streams * (2 if cfg['flip'] else 1)
    dev = torch.device('cuda:0')
    model = DLASegHIP(heads)
    model.load_state_dict(calibrated_state_dict(name, heads))
    model = model.to(dev).eval()
    plan = model.get_plan(N, H, W, True, True, True)
    bufs = plan_buffers(plan)
    xi, ii, hi = plan['inputs']
    ins = []
    for j in range(inputs):
        g = torch.Generator().manual_seed(900 + j)
        ins.append((torch.randn((N, 3, H, W), generator=g).to(dev), torch.randn((N, 3, H, W), generator=g).to(dev),
                    (torch.rand((N, 1, H, W), generator=g) ** 8).to(dev)))
    graph = None
    if use_graph:
        model._run_plan(plan)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            model._run_plan(plan)
    refs = [None] * inputs
    events = 0
    for r in range(runs):
        j = r % inputs
        xi.copy_(ins[j][0]); ii.copy_(ins[j][1]); hi.copy_(ins[j][2])
        if graph is not None:
            graph.replay()
        else:
            model._run_plan(plan)
        torch.cuda.synchronize()
        if refs[j] is None:
            refs[j] = [t.clone() for _, _, _, t in bufs]
            continue
        outs_equal = all(torch.equal(t, refs[j][b]) for b, (_, _, _, t) in enumerate(bufs))
        if outs_equal:
            continue
        events += 1
        bad = []
        for b, (i, lname, k, t) in enumerate(bufs):
            if not torch.equal(t, refs[j][b]):
                d = (t.double() - refs[j][b].double()).cpu().numpy()
                bad.append({'launch': i, 'name': lname, 'k': k, 'shape': list(t.shape), 'n': int((d != 0).sum()),
                            'max': float(np.nanmax(np.abs(d))), 'box': where(d)})
        detail = None
        for b, (i, lname, k, t) in enumerate(bufs):
            if not torch.equal(t, refs[j][b]):
                flat, ref = t.flatten(), refs[j][b].flatten()
                idx = torch.nonzero(flat != ref).flatten()[:64]
                detail = {'name': lname, 'k': k, 'shape': list(t.shape), 'idx': idx.tolist(), 'got': flat[idx].tolist(),
                          'want': ref[idx].tolist(),
                          'other_input_ref': [refs[o][b].flatten()[idx].tolist() for o in range(inputs) if o != j and refs[o] is not None]}
                break
        print(json.dumps({'config': name, 'streams': streams, 'graph': bool(use_graph), 'run': r, 'input': j,
                          'first': detail, 'differing_buffers': bad[:6], 'n_differing': len(bad)}), flush=True)
    print(json.dumps({'config': name, 'streams': streams, 'graph': bool(use_graph), 'runs': runs, 'events': events,
                      'launches': [l.name for l in plan['launches']]}), flush=True)
    return events


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', type=int, default=0, help='forward-only mode: this many runs per configuration')
    ap.add_argument('--graph', type=int, default=0)
    ap.add_argument('--config', action='append')
    ap.add_argument('--streams', type=int, action='append')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--passes', type=int, default=4)
    ap.add_argument('--poison', type=float, default=8.0, help='GiB of allocator pool filled with NaNs between passes')
    ap.add_argument('--no-heads', action='store_true')
    ap.add_argument('--sparse-heads', action='store_true', help='stream passes with opt.sparse_heads (the opt-in mode)')
    a = ap.parse_args()
    cfgs = a.config or ['coco_512']
    strs = a.streams or [4] * len(cfgs)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    rc = 0
    for name, streams in zip(cfgs, strs):
        if a.model:
            rc |= 1 if model_mode(name, streams, a.model, a.graph) else 0
            continue
        first = one_pass(name, streams, a.frames, 317 + 7, not a.no_heads, sparse_heads=a.sparse_heads)
        for p in range(1, a.passes):
            poison_free_memory(dev, a.poison)
            cur = one_pass(name, streams, a.frames, 317 + 7, not a.no_heads, sparse_heads=a.sparse_heads)
            d = diff_passes(first, cur)
            line = {'config': name, 'streams': streams, 'frames': a.frames, 'pass': p, 'identical': d is None}
            if d is not None:
                rc = 1
                line.update(first_diff_frame=d[0], diff=d[1])
            print(json.dumps(line), flush=True)
    return rc


if __name__ == '__main__':
    sys.exit(main())
