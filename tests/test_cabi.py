"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/centertrack_hip.h declares (no compute calls without a GPU); host-side argument
validation returns error codes instead of crashing."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def lib():
    from centertrack_amd import build
    path = build.build()
    assert os.path.exists(path)
    return ctypes.CDLL(path)


def _declared():
    src = open(os.path.join(ROOT, 'include', 'centertrack_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ct_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n
    from centertrack_amd import _lib
    assert sorted(_lib.EXPORTS) == names, 'python binding list out of sync with the header'


def test_version_and_errors_without_gpu(lib):
    from centertrack_amd import _lib
    l = _lib.load()
    assert l.ct_version() == _lib.ABI_VERSION == 103
    d = _lib.ConvDesc()
    assert l.ct_conv2d(ctypes.byref(d), None) == 1          # CT_ERR_ARG: null pointers
    assert b'null' in l.ct_last_error()
    assert l.ct_packed_weight_elems(27, 64, 3) == 9 * 4 * 2 * 256
    dd = _lib.DecodeDesc()
    assert l.ct_decode(ctypes.byref(dd), None) == 1


def test_product_never_imports_oracle():
    """the product package must not route through the CPU oracle"""
    pkg = os.path.join(ROOT, 'centertrack_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_header_is_plain_c99(tmp_path):
    """the drop-in boundary is a C ABI: include/centertrack_hip.h must compile as C99 (no C++-isms, no torch types)"""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "centertrack_hip.h"\nint main(void) { ct_conv_desc c; ct_dcn_desc d; ct_decode_desc e; '
                   'ct_pose_desc p; (void)c; (void)d; (void)e; (void)p; return ct_version() > 0 ? 0 : 1; }\n')
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(root, 'include'),
                        '-fsyntax-only', str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_descriptors_match_the_header_layout(tmp_path):
    """the ctypes mirrors of the descriptors (centertrack_amd/_lib.py) must have the size and the field offsets a C
    compiler gives the structs of include/centertrack_hip.h (ct_dcn_desc grew in round 2: om_partial)"""
    import shutil
    import subprocess
    from centertrack_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    checks = {'ct_dcn_desc': (_lib.DcnDesc, ['x', 'om', 'w_packed', 'workspace', 'fuse_offset', 'w_off_packed', 'up_w',
                                             'up_ldy', 'om_partial', 'om_partial_bytes']),
              'ct_conv_desc': (_lib.ConvDesc, ['x', 'split_k', 'algo', 'w_winograd', 'pool_y', 'pool_ld']),
              'ct_pose_desc': (_lib.PoseDesc, ['rows', 'box_col', 'hp_offset', 'out', 'workspace_bytes', 'box_wh', 'box_ltrb',
                                               'box_ltrb_batch_stride']),
              'ct_decode_desc': (_lib.DecodeDesc, ['hm', 'heads', 'out', 'hm_batch_stride', 'out_stride', 'host_out', 'done_flag',
                                                   'done_counter', 'sparse']),
              'ct_sparse_heads_desc': (_lib.SparseHeadsDesc, ['feat', 'ldf', 'nheads', 'head', 'w1', 'b1', 'w2', 'b2',
                                                              'depth_scale', 'zero_tracking', 'flip_B', 'flip_mode']),
              'ct_heads_desc': (_lib.HeadsDesc, ['x', 'w0_winograd', 'cout', 'out', 'depth_scale']),
              'ct_frame_loop_desc': (_lib.FrameLoopDesc, ['B', 'trackers', 'layout', 'out_thresh', 'host_rows', 'rows_keep',
                                                          'blob_params', 'blob_cap', 'nslots', 'graphs', 'frames',
                                                          'frame_bytes', 'stream', 'results', 'results_cap', 'done_flag', 'pre']),
              'ct_prestage_desc': (_lib.PrestageDesc, ['enabled', 'W', 'w_x', 'shift3', 'partial', 'ldp', 'flip_B']),
              'ct_frame_step_args': (_lib.FrameStepArgs, ['slot', 'frame_kind', 'frame', 'next_frame', 'trans_input',
                                                          'trans_inv']),
              'ct_track': (_lib.Track, ['score', 'ct', 'bbox', 'tracking_id', 'row'])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "centertrack_hip.h"', 'int main(void) {']
    for cname, (_, fields) in checks.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ['return 0; }']
    src = tmp_path / 'lay.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'lay'
    r = subprocess.run([gcc, '-std=c99', '-I' + os.path.join(ROOT, 'include'), str(src), '-o', str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for cname, (cls, fields) in checks.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in fields:
            assert int(got['%s.%s' % (cname, f)]) == getattr(cls, f).offset, '%s.%s' % (cname, f)


def test_dcn_argument_validation_without_gpu(lib):
    """make_plan rejects bad fuse_offset modes / missing buffers before anything is launched"""
    from centertrack_amd import _lib
    l = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    d = _lib.DcnDesc()
    d.x = d.w_packed = d.y = d.om = p
    d.N, d.H, d.W, d.Cin, d.Cout, d.ldx, d.ldy, d.ldom = 1, 4, 4, 64, 64, 64, 64, 32
    d.fuse_offset = 4
    assert l.ct_dcn_v2(ctypes.byref(d), None) == _lib.CT_ERR_ARG and b'fuse_offset' in l.ct_last_error()
    d.fuse_offset = 2                                          # K-split offsets without weights / bias
    assert l.ct_dcn_v2(ctypes.byref(d), None) == _lib.CT_ERR_ARG
    d.w_off_packed = d.b_off = p
    assert l.ct_dcn_v2_offsets_bytes(ctypes.byref(d)) == 1 * 4 * 4 * 32 * 4
    assert l.ct_dcn_v2(ctypes.byref(d), None) == _lib.CT_ERR_WORKSPACE      # no om_partial
    d.fuse_offset = 3
    assert l.ct_dcn_v2_offsets_bytes(ctypes.byref(d)) == 1 * 4 * 4 * 32 * 4
    d.Cin = 96
    d.fuse_offset = 1
    assert l.ct_dcn_v2(ctypes.byref(d), None) == _lib.CT_ERR_ARG            # fused offsets need Cin % 64 == 0
    assert l.ct_dcn_v2_group(ctypes.byref(d), 1, 0, None) == _lib.CT_ERR_ARG  # no phase named
    assert l.ct_dcn_v2_group(ctypes.byref(d), 5, _lib.CT_DCN_MAIN, None) == _lib.CT_ERR_ARG


def test_group_plan_query_reports_errors_instead_of_zero(lib):
    """ct_dcn_v2_group_workspace_bytes answers 0 for "none needed" AND for a rejected descriptor; ct_dcn_v2_group_plan
    tells them apart"""
    from centertrack_amd import _lib
    l = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    d = _lib.DcnDesc()
    d.x = d.w_packed = d.om = p
    d.N, d.H, d.W, d.Cin, d.Cout, d.ldx, d.ldy, d.ldom = 1, 8, 8, 256, 64, 256, 64, 32
    d.algo, d.split_k = 3264, 2
    need, splits = ctypes.c_size_t(7), ctypes.c_int(0)
    assert l.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), ctypes.byref(splits)) == 0
    assert splits.value == 2 and need.value == 2 * 8 * 8 * 64 * 4 == l.ct_dcn_v2_group_workspace_bytes(ctypes.byref(d))
    d.split_k = 1
    assert l.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), None) == 0 and need.value == 0
    # the 64-channel-step shape resolves split counts in units of 64 channels
    d.algo, d.split_k = 43264, 8                               # 256 channels = 4 units of 64: at most 4 splits
    assert l.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), ctypes.byref(splits)) == 0 and splits.value == 4
    d.algo = 41664                                             # (an experimental shape of round 3, removed in round 4)
    assert l.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), None) == _lib.CT_ERR_ARG and b'unknown algo' in l.ct_last_error()
    d.Cout, d.ldy, d.algo, d.split_k = 64, 64, 3264, 1
    d.Cin = 48                                                 # rejected: the size query says 0, the plan query says why
    assert l.ct_dcn_v2_group_workspace_bytes(ctypes.byref(d)) == 0
    assert l.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), None) == _lib.CT_ERR_ARG
    assert b'Cin' in l.ct_last_error()


def test_round3_entry_points_validate_their_arguments_without_gpu(lib):
    """ct_frame_loop_create / ct_signal_host / ct_stem_forward_parts / ct_decode (host rows) reject bad descriptors before
    anything touches a device"""
    from centertrack_amd import _lib
    l = _lib.load()
    assert not l.ct_frame_loop_create(None) and b'bad descriptor' in l.ct_last_error()
    d = _lib.FrameLoopDesc()
    d.B, d.K, d.F = 1, 100, 17
    assert not l.ct_frame_loop_create(ctypes.byref(d)) and b'bad descriptor' in l.ct_last_error()     # no trackers / rows / results
    assert l.ct_frame_loop_submit(None, None) == _lib.CT_ERR_ARG
    assert l.ct_frame_loop_wait(None) == _lib.CT_ERR_ARG
    assert l.ct_frame_loop_pending_slot(None) == -1 and l.ct_frame_loop_in_flight(None) == -1
    l.ct_frame_loop_destroy(None)                                 # a no-op
    assert l.ct_signal_host(None, 1, None) == _lib.CT_ERR_ARG
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # no stem selected / input without weights
    assert l.ct_stem_forward_parts(None, None, None, None, 0, 1, 32, 32, None, None, None, p, p, p, 16, None) == _lib.CT_ERR_ARG
    assert b'no stem' in l.ct_last_error()
    assert l.ct_stem_forward_parts(p, None, None, None, 0, 1, 32, 32, None, None, None, p, p, p, 16, None) == _lib.CT_ERR_ARG
    dd = _lib.DecodeDesc()
    dd.hm, dd.out, dd.B, dd.C, dd.h, dd.w, dd.K = p, p, 1, 1, 16, 16, 10
    need = l.ct_decode_workspace_bytes(ctypes.byref(dd))
    assert need > 0
    dd.workspace, dd.workspace_bytes = p, need
    dd.done_flag = p                                              # flag without host rows / arrival counter
    assert l.ct_decode(ctypes.byref(dd), None) == _lib.CT_ERR_ARG and b'done_flag' in l.ct_last_error()


def test_pinned_table_wins_over_a_user_cache_and_is_never_copied_into_it(tmp_path, monkeypatch):
    """a CENTERTRACK_TUNE_CACHE file written before the package shipped a re-tuned pinned table must not
    resurrect stale shapes, and the file only ever holds keys the pinned table does not"""
    import json
    from centertrack_amd import autotune
    with open(autotune.PINNED_TABLE) as f:
        pinned = json.load(f)
    key = sorted(k for k in pinned if k.startswith('dcnplan'))[0]
    stale = {key: [0, 2, 2, 1, 999.0], 'conv:user-only-key': [3, 1, 5.0]}
    cache = tmp_path / 'cache.json'
    cache.write_text(json.dumps(stale))
    monkeypatch.setenv('CENTERTRACK_TUNE_CACHE', str(cache))
    monkeypatch.delenv('RANK', raising=False)
    saved = (dict(autotune._CACHE), dict(autotune._PINNED), autotune._LOADED)
    try:
        autotune._CACHE.clear(); autotune._PINNED.clear(); autotune._LOADED = False
        autotune._load_file()
        assert list(autotune._CACHE[key]) == pinned[key]                  # the pinned entry won
        assert list(autotune._CACHE['conv:user-only-key']) == [3, 1, 5.0]   # other keys of the file are honoured
        autotune._CACHE['conv:live-tuned'] = (2, 1, 7.5)
        autotune._save_file()
        written = json.loads(cache.read_text())
        assert set(written) == {'conv:user-only-key', 'conv:live-tuned'}
    finally:
        autotune._CACHE.clear(); autotune._CACHE.update(saved[0])
        autotune._PINNED.clear(); autotune._PINNED.update(saved[1])
        autotune._LOADED = saved[2]


def test_pinned_tune_table_is_well_formed():
    """centertrack_amd/tune_table.json: conv entries (algo, split_k, us), DCN schedule entries `dcnplan3:N,H,W` ->
    (fuse_max_cin, chunks_per_split, nkk, offset mode, us) with values the scheduler understands; no stale keys"""
    import json
    from centertrack_amd import autotune
    with open(autotune.PINNED_TABLE) as f:
        t = json.load(f)
    plans = {k: v for k, v in t.items() if k.startswith('dcnplan')}
    assert len(plans) >= 20 and all(k.startswith(('dcnplan3:', 'dcnplan4:', 'dcnplan5:')) for k in plans)
    for k, v in plans.items():
        n, h, w = (int(x) for x in k.split(':')[1].split(','))
        assert n >= 1 and h % 32 == 0 and w % 32 == 0
        # (round 3 added the two knobs of the small slots, round 6 the persistent MAIN launches)
        nk = 4 if k.startswith('dcnplan3:') else 6 if k.startswith('dcnplan4:') else 7
        assert len(v) == nk + 1 and v[0] in (0, 64, 128, 256) and v[1] in (2, 4, 8) and v[2] in (2, 4) and v[3] in (1, 2, 3), (k, v)
        assert nk == 4 or (v[4] in (0, 1, 2) and v[5] in (0, 1, 2)), (k, v)
        assert nk < 7 or v[6] in (0, 1), (k, v)
        assert v[nk] > 0
    for k, v in t.items():
        if k.startswith('conv'):
            assert len(v) == 3 and v[2] > 0, (k, v)
    for must in ('1,512,512', '8,384,1280', '4,512,512', '4,448,800'):
        assert 'dcnplan3:' + must in t or 'dcnplan4:' + must in t, must          # BASELINE configs 2 / 3 / 4 / 5


def test_tools_compile():
    import py_compile
    tools = os.path.join(ROOT, 'tools')
    for dirpath, _, files in os.walk(tools):
        for fn in files:
            if fn.endswith('.py'):
                py_compile.compile(os.path.join(dirpath, fn), doraise=True)
    py_compile.compile(os.path.join(ROOT, 'bench.py'), doraise=True)
    py_compile.compile(os.path.join(ROOT, '__graft_entry__.py'), doraise=True)
