"""CPU: the pure-Python pieces of the launch plan (no device needed): the MAIN-slot structure of the 16 DCN nodes of
DLAUp / IDAUp (dla.py:539-574) as `_dcn_slot_sizes` sees it, and the plan signature ranks compare."""
import types

from centertrack_amd.model import DLASegHIP, _DcnLayer


class _Buf(object):
    pass


def _view(H, W, C):
    return types.SimpleNamespace(buf=_Buf(), H=H, W=W, C=C)


def _dla_up_layers():
    """the 16 DeformConv nodes in the reference's order, with the dataflow of DLAUp.forward / IDAUp.forward at 512x512"""
    feats = [_view(512, 512, 16), _view(256, 256, 32), _view(128, 128, 64), _view(64, 64, 128), _view(32, 32, 256), _view(16, 16, 512)]
    layers_out = []

    def ida(name, layers, startp, endp, o, up_f):
        for i in range(startp + 1, endp):
            k = i - startp
            f = up_f[k]
            xi = layers[i]
            up = _view(xi.H * f, xi.W * f, o)
            layers_out.append(_DcnLayer('%s.proj_%d' % (name, k), xi, o, _view(xi.H, xi.W, o), (None, f, layers[i - 1], up)))
            node_out = _view(up.H, up.W, o)
            layers_out.append(_DcnLayer('%s.node_%d' % (name, k), up, o, node_out, None))
            layers[i] = node_out

    layers = list(feats)
    outs = [layers[-1]]
    ida('dla_up.ida_0', layers, 4, 6, 256, [1, 2]); outs.insert(0, layers[-1])
    ida('dla_up.ida_1', layers, 3, 6, 128, [1, 2, 2]); outs.insert(0, layers[-1])
    ida('dla_up.ida_2', layers, 2, 6, 64, [1, 2, 2, 2]); outs.insert(0, layers[-1])
    y = [outs[0], outs[1], outs[2]]
    ida('ida_up', y, 0, 3, 64, [1, 2, 4])
    return feats, layers_out


def test_dcn_main_slots_of_one_stream_at_512():
    """8 dependent MAIN slots; the three `proj` slots are the small ones (384 / 320 / 256 workgroups
    of 768 resident), the first `node` slot the largest"""
    feats, layers = _dla_up_layers()
    produced0 = {id(f.buf): 0 for f in feats}
    sizes, mains = DLASegHIP._dcn_slot_sizes(layers, produced0, 1, 4, with_mains=True)
    assert sorted(sizes) == [1, 2, 3, 4, 5, 6, 7, 8]
    assert sizes == {1: 384, 2: 1024, 3: 320, 4: 768, 5: 256, 6: 512, 7: 512, 8: 512}
    by_name = dict(zip([ly.name for ly in layers], mains))
    assert by_name['dla_up.ida_0.proj_1'] == by_name['dla_up.ida_1.proj_1'] == by_name['dla_up.ida_2.proj_1'] == 1
    assert by_name['ida_up.proj_2'] == 3 and by_name['ida_up.proj_1'] == 5
    assert [by_name['dla_up.ida_2.node_3'], by_name['ida_up.node_1'], by_name['ida_up.node_2']] == [6, 7, 8]
    # the slot of a layer does not depend on how finely its producer is split ...
    for cps in (2, 8):
        assert DLASegHIP._dcn_slot_sizes(layers, produced0, 1, cps, with_mains=True)[1] == mains
    # ... except through a skip input: with the 64-channel nodes split too, ida_up.proj_1's finishing step waits one
    # slot longer for dla_up.ida_2.node_3 and the last two nodes move down
    m1 = DLASegHIP._dcn_slot_sizes(layers, produced0, 1, 1, with_mains=True)[1]
    assert m1[:13] == mains[:13] and m1[13] == 8 and m1[15] == 10
    # two chunks per split doubles the workgroups of every layer with >= 128 input channels
    s2 = DLASegHIP._dcn_slot_sizes(layers, produced0, 1, 2)
    assert s2[1] == 768 and s2[3] == 640 and s2[5] == 512 and s2[6] == 512


def test_plan_signature_pins_algos_splits_and_knobs():
    conv = types.SimpleNamespace(algo=203, split_k=1)
    la = types.SimpleNamespace(fn='conv', name='level0', args=conv)
    arr = [types.SimpleNamespace(algo=43264, split_k=2, fuse_offset=2)]
    lb = types.SimpleNamespace(fn='dcn_group', name='dcn[x]', args=(arr, 1, 1))
    lc = types.SimpleNamespace(fn='heads', name='heads.fused', args=None)
    plan = {'launches': [la, lb, lc], 'dcn_knobs': (128, 4, 4, 1, 0, 0)}
    sig = DLASegHIP.plan_signature(plan)
    assert 'level0:conv:203:1' in sig and 'dcn[x]:dcn:1:43264/2/2' in sig and 'dcn_knobs=(128, 4, 4, 1, 0, 0)' in sig
    conv.algo = 204
    assert DLASegHIP.plan_signature(plan) != sig
