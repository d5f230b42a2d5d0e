"""The shipped device code holds no wide store whose data registers the next instructions overwrite (tools/isa_hazards.py).

Round 6: the Winograd OFFSETS kernel stored the NEXT instruction's result instead of its own in four lanes of every 16, once
per ~15 launches -- `buffer_store_dwordx4` with a scalar-offset register followed directly by a VALU write of its first data
register: hipcc pads that pair only when the store has no scalar-offset register (the published exemption), gfx950 needs the
pad either way.  Found by tests/test_hip_determinism.py's bitwise run-to-run comparison; this test keeps the pattern out of
every kernel at build time, without a GPU."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_lint_recognises_the_pattern():
    import isa_hazards as H
    listing = [('k', ) + H.parse(l) + (l,) for l in (
        'buffer_store_dwordx4 v[2:5], v23, s[8:11], s19 offen',
        'v_or_b32_e32 v2, 2, v11',
    )]
    mn, ops = listing[0][1], listing[0][2]
    assert H.WIDE_STORE.match(mn) and H.store_data(mn, ops) == {2, 3, 4, 5}
    assert H.valu_writes(listing[1][1], listing[1][2]) == {2}
    assert H.store_data(*H.parse('global_store_dwordx4 v[8:9], v[4:7], off')) == {4, 5, 6, 7}
    assert H.valu_writes(*H.parse('v_cmp_gt_i32_e32 vcc, s22, v2')) == set()
    assert H.valu_writes(*H.parse('v_mad_u64_u32 v[2:3], s[8:9], s23, v20, v[126:127]')) == {2, 3}
    assert not H.WIDE_STORE.match('buffer_store_dwordx2')


def test_no_wide_store_is_overwritten_within_two_issue_slots():
    import isa_hazards as H
    from centertrack_amd import build
    build.build()                      # (no-op when the objects are fresh)
    n, findings = H.run()
    assert n >= 9, 'expected one gfx950 code object per .hip translation unit, found %d' % n
    assert not findings, 'wide stores followed by a VALU write of their data registers:\n%s' % '\n'.join(
        '%s %s: %s -> %s' % f for f in findings)
