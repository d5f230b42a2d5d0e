"""Per-layer launch-shape selection by measurement ("measure, don't guess").

Every dense conv / DCNv2 launch of a plan can run with several tile shapes (row-tiled
workgroups of 256px x 16 .. 32px x 64 couts with optional split-K, or the K-split-in-
workgroup kernel, see csrc/conv_mfma.hip; 64 / 128 couts per workgroup and split-K for
csrc/dcn_mfma.hip).  Which one is fastest depends on (map size, channels, batch) in ways a
static rule gets wrong at the extremes (16x16 maps with 1280 channels, 27-cout offset convs),
so the first time a (shape) key is seen its candidates are timed on the device -- a HIP graph
of back-to-back launches of the real descriptor, i.e. device time incl. the kernel boundary,
free of host launch cost -- and the winner is cached for the process (optionally in a JSON
file named by ``CENTERTRACK_TUNE_CACHE`` so that a deployment replays identical choices: the
choice fixes the fp32 summation order, hence the last bits of the results).

``CENTERTRACK_AUTOTUNE=0`` keeps the built-in heuristics (``algo = 0``).
"""
import ctypes
import json
import os

import torch

from . import _lib

_CACHE = {}
_PINNED = {}
_LOADED = False
_SCRATCH = {}
_SHADOW = {}
REPS = 12


def enabled():
    return os.environ.get('CENTERTRACK_AUTOTUNE', '1') != '0'


def _cache_path():
    return os.environ.get('CENTERTRACK_TUNE_CACHE', '')


# (CENTERTRACK_TUNE_TABLE: another pinned table, for A/B runs of a re-tuned one before it replaces the shipped file)
PINNED_TABLE = os.environ.get('CENTERTRACK_TUNE_TABLE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tune_table.json')


def _read_table(path):
    """{key: (algo, split_k, us)} of a table file; a missing, truncated or otherwise unreadable file is ignored
    (its keys are simply tuned again) instead of failing the start-up"""
    try:
        with open(path) as f:
            return {k: tuple(v) for k, v in json.load(f).items()}
    except (OSError, ValueError, TypeError):
        return {}


def _load_file():
    """Choices are taken, in this order, from (1) the pinned table shipped in the package (``tune_table.json``:
    the launch shapes of the BASELINE.json configurations, measured once on an MI355X and committed, so that every
    process -- every rank of a torchrun job -- replays IDENTICAL tile shapes, hence identical fp32 summation orders
    and identical last bits; ``CENTERTRACK_TUNE_PINNED=0`` ignores it), (2) the file named by
    ``CENTERTRACK_TUNE_CACHE`` -- for keys the pinned table does not hold: on a conflict the PINNED entry wins, so a
    user cache written before the package shipped a re-tuned table cannot resurrect stale shapes -- (3) live timing."""
    global _LOADED
    if _LOADED:
        return
    _LOADED = True
    p = _cache_path()
    if p:
        _CACHE.update(_read_table(p))
    if os.environ.get('CENTERTRACK_TUNE_PINNED', '1') != '0':
        _PINNED.update(_read_table(PINNED_TABLE))
        _CACHE.update(_PINNED)


def _save_file():
    """atomic (temp file + rename) and written by rank 0 only: under torchrun every rank tunes the same keys, and
    concurrent rewrites of one path would interleave.  Only the keys the pinned table does NOT hold are written: the
    file never carries a copy of pinned entries that could outlive a re-tuned table."""
    p = _cache_path()
    if not p or int(os.environ.get('RANK', '0')) != 0:
        return
    tmp = '%s.tmp.%d' % (p, os.getpid())
    with open(tmp, 'w') as f:
        json.dump({k: list(v) for k, v in sorted(_CACHE.items()) if k not in _PINNED}, f, indent=0)
    os.replace(tmp, p)


def share_from_rank0():
    """torchrun jobs: a key missing from the pinned table / cache file is timed live, and two ranks timing the same
    candidates may pick different winners (other fp32 summation orders).  Rank 0's choices are therefore broadcast and
    adopted by every rank: call it after rank 0 built its plans and before the other ranks build theirs
    (``parallel.build_plans_consistently``).  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    _load_file()
    box = [{k: list(v) for k, v in _CACHE.items()} if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if dist.get_rank() != 0:
        _CACHE.update({k: tuple(v) for k, v in box[0].items()})


def _scratch(nbytes, device):
    key = str(device)
    t = _SCRATCH.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty(max(nbytes, 1 << 20) // 4 + 4, dtype=torch.float32, device=device)
        _SCRATCH[key] = t
    return t


def cold_inputs():
    """``CENTERTRACK_TUNE_COLD=1``: every timed launch is preceded by a device copy that re-writes its input
    (from a shadow buffer), so that the candidate finds its input where a real frame leaves it -- just written by
    another kernel, i.e. dirty in the producer XCDs' L2 / in the Infinity Cache -- instead of hot in its own L2 after
    the previous repetition.  The copy's own time (measured alone) is subtracted."""
    return os.environ.get('CENTERTRACK_TUNE_COLD', '0') == '1'


def _time_graph(fn, reps=REPS, pre=None):
    """us per launch of ``fn`` (a C-ABI call on the current stream), via graph replay; ``pre``: optional call
    enqueued before every launch (its cost is measured separately and subtracted)."""
    if pre is not None:
        both = _time_graph(lambda: (pre(), fn())[1], reps)
        if both is None:
            return None
        alone = _time_graph(lambda: pre() or 0, reps)
        return max(both - (alone or 0.0), 0.01)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        if fn() != 0:
            torch.cuda.current_stream().wait_stream(side)
            return None
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _lib.capture_guard(collect=False), torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(2):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / reps
        best = t if best is None else min(best, t)
    del g
    return best


def _conv_key(d):
    # ('P': the launch also computes the fused projection -- other register / LDS use, other winners)
    return 'conv%s%s:%d,%d,%d,%d,%d,%d,%d,%d,%d' % ('W' if d.w_winograd else '', 'P' if d.proj_w_packed else '', d.N, d.H, d.W, d.Cin, d.Cout, d.ks,
                                                   d.stride, 1 if d.res else 0, 1 if (d.flags & _lib.CT_OUT_NCHW) else 0)


def _dcn_key(d):
    return 'dcn%s%s:%d,%d,%d,%d,%d' % ('F' if d.fuse_offset else '', 'U%d' % d.up_f if d.up_w else '', d.N, d.H, d.W,
                                       d.Cin, d.Cout)


_REG_BN = [16, 32, 64, 128, 64, 32, 16, 16]  # couts per workgroup of the row-tiled shapes 0..7
_KS = [(2, 2, 4), (1, 2, 4), (1, 2, 8), (2, 4, 4), (2, 2, 8)]


def _conv_candidates(d):
    cout_pad = (d.Cout + 15) // 16 * 16
    cands = [(0, 0)]
    for cfg, bn in enumerate(_REG_BN):
        if bn > max(32, cout_pad) or (cout_pad + bn - 1) // bn > 24 or cfg == 3:
            continue
        cands.append((cfg + 1, 0))
        cands.append((cfg + 1, 1))
    for i, (wm, wn, wk) in enumerate(_KS):
        if d.Cin % (16 * wk) or (d.stride == 2 and wm == 2 and wk == 8):
            continue
        if 16 * wn > max(32, cout_pad):
            continue
        cands.append((101 + i, 1))
    if d.w_winograd and d.ks == 3 and d.stride == 1 and d.Cin % 64 == 0 and not (d.flags & _lib.CT_OUT_NCHW):
        for algo in (201, 202, 203, 204, 205, 206, 207):   # Winograd F(2x2,3x3) tile / K-split shapes (centertrack_hip.h)
            cands.append((algo, 1))
        if d.Cin == 64 and d.Cout >= 128:                   # 2 / 4 / 5 / 8 cout blocks per workgroup on one input transform
            cands += [(208, 1), (209, 1), (210, 1), (211, 1)]
    return cands


def _dcn_candidates(d):
    cands = [] if d.fuse_offset else [(0, 0)]
    nchunks = d.Cin // 32
    algos = [3264, 32128] if d.fuse_offset else [64, 128, 3264, 32128]
    for bn in (algos if d.Cout >= 128 else [a for a in algos if a in (64, 3264)]):
        for sk in (1, 2, 4, 8, 16):
            if sk <= nchunks:
                cands.append((bn, sk))
    return cands


def _tune(d, key, cands, call, ws_bytes_fn, device):
    _load_file()
    if key in _CACHE:
        return _CACHE[key]
    saved = (d.algo, d.split_k, d.workspace, d.workspace_bytes)
    pre = None
    if cold_inputs():
        # (the input may be a channel slice of a wider NHWC buffer: stop at the last pixel's last channel)
        nbytes = ((int(d.N) * int(d.H) * int(d.W) - 1) * int(d.ldx) + int(d.Cin)) * 4
        shadow = _SHADOW.get(str(device))
        if shadow is None or shadow.numel() * 4 < nbytes:
            shadow = torch.zeros(max(nbytes, 1 << 22) // 4, dtype=torch.float32, device=device)
            _SHADOW[str(device)] = shadow
        lib = _lib.load()
        # (shadow <- input once, then input <- shadow before every timed launch: same values, freshly written)
        _lib.check(lib.ct_memcpy_async(shadow.data_ptr(), d.x, nbytes, 0, _lib.stream_ptr()), 'ct_memcpy_async')
        torch.cuda.synchronize()
        x_ptr, sh_ptr = int(d.x), shadow.data_ptr()
        pre = lambda: lib.ct_memcpy_async(x_ptr, sh_ptr, nbytes, 0, _lib.stream_ptr())
    results = []
    for algo, sk in cands:
        d.algo, d.split_k = algo, sk
        d.workspace, d.workspace_bytes = None, 0
        need = ws_bytes_fn(ctypes.byref(d))
        ws = _scratch(need, device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        t = _time_graph(lambda: call(ctypes.byref(d), _lib.stream_ptr()), pre=pre)
        if t is not None:
            results.append((t, algo, sk))
    d.algo, d.split_k, d.workspace, d.workspace_bytes = saved
    if not results:
        best = (0, 0, 0.0)
    else:
        results.sort()
        best = (results[0][1], results[0][2], round(results[0][0], 2))
    _CACHE[key] = best
    _save_file()
    return best


def tune_conv(d, device):
    """Pick (algo, split_k) for a ct_conv_desc; sets them on ``d``; returns (algo, split_k, us)."""
    lib = _lib.load()
    algo, sk, us = _tune(d, _conv_key(d), _conv_candidates(d), lib.ct_conv2d, lib.ct_conv2d_workspace_bytes, device)
    d.algo, d.split_k = algo, sk
    return algo, sk, us


def tune_dcn(d, device):
    lib = _lib.load()
    algo, sk, us = _tune(d, _dcn_key(d), _dcn_candidates(d), lib.ct_dcn_v2, lib.ct_dcn_v2_workspace_bytes, device)
    d.algo, d.split_k = algo, sk
    return algo, sk, us


def report():
    """{key: (algo, split_k, us)} of everything tuned so far (for DESIGN.md / debugging)."""
    return dict(_CACHE)
