"""GPU parity of the drop-in op boundaries: B1 ``DCN`` / ``DCNv2`` /
``dcn_v2_conv`` with upstream's signatures and state-dict keys, and B3 ``generic_decode``.
The checker is the CPU oracle (oracle/dcn_v2.py, oracle/decode.py); tolerance for the fp32
contraction with a different summation order: 2e-4 abs (north_star allows 1e-3); decode
indices / classes / gathered values bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).float()


@pytest.mark.parametrize('cin,cout,shape', [(64, 64, (1, 16, 24)), (128, 64, (2, 9, 21)), (256, 128, (1, 8, 8))])
def test_DCN_module_is_upstream_DCN(device, cin, cout, shape):
    from centertrack_amd import dcn_v2 as hip
    from oracle import dcn_v2 as odcn
    ref = odcn.DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    ref.conv_offset_mask.weight.data = _rand(27, cin, 3, 3, seed=1, scale=0.03)
    ref.conv_offset_mask.bias.data = _rand(27, seed=2, scale=0.3)
    ref.bias.data = _rand(cout, seed=3)
    mod = hip.DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    assert list(mod.state_dict().keys()) == ['weight', 'bias', 'conv_offset_mask.weight', 'conv_offset_mask.bias']
    assert float(mod.conv_offset_mask.weight.abs().max()) == 0.0          # upstream init_offset()
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(device)
    x = torch.relu(_rand(shape[0], cin, shape[1], shape[2], seed=4))
    with torch.no_grad():
        want = ref(x)
    got = mod(x.to(device))
    assert got.shape == want.shape and got.is_contiguous()
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-4, rtol=1e-4)
    # weights are re-packed when a parameter changes in place
    with torch.no_grad():
        mod.weight.mul_(2.0)
        mod.bias.zero_()
        ref.weight.mul_(2.0)
        ref.bias.zero_()
        want2 = ref(x)
    np.testing.assert_allclose(mod(x.to(device)).cpu().numpy(), want2.numpy(), atol=4e-4, rtol=1e-4)


def test_DCNv2_and_functional(device):
    from centertrack_amd import dcn_v2 as hip
    from oracle import dcn_v2 as odcn
    x = _rand(2, 64, 11, 13, seed=5)
    off = _rand(2, 18, 11, 13, seed=6, scale=2.0)
    mask = torch.sigmoid(_rand(2, 9, 11, 13, seed=7))
    w, b = _rand(32, 64, 3, 3, seed=8, scale=1 / 24.), _rand(32, seed=9)
    want = odcn.dcn_v2_conv(x, off, mask, w, b)
    got = hip.dcn_v2_conv(x.to(device), off.to(device), mask.to(device), w.to(device), b.to(device), 1, 1, 1, 1)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-4, rtol=1e-4)
    mod = hip.DCNv2(64, 32, (3, 3), 1, 1).to(device)
    with torch.no_grad():
        mod.weight.copy_(w)
        mod.bias.copy_(b)
    got = mod(x.to(device), off.to(device), mask.to(device))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-4, rtol=1e-4)


def test_DCN_rejects_what_it_does_not_implement(device):
    from centertrack_amd import _lib, dcn_v2 as hip
    with pytest.raises(_lib.CTError):
        hip.DCN(64, 64, (3, 3), 2, 1)                    # stride 2
    with pytest.raises(_lib.CTError):
        hip.DCN(48, 64, (3, 3), 1, 1)                    # Cin % 32
    mod = hip.DCN(64, 64, (3, 3), 1, 1)
    with pytest.raises(_lib.CTError):
        mod(torch.zeros(1, 64, 8, 8))                    # CPU tensor: no fallback


def test_generic_decode_dropin(device):
    from types import SimpleNamespace
    import scenarios as S
    from centertrack_amd.decode import generic_decode
    from oracle import decode as odecode
    assert generic_decode({'reg': torch.zeros(1, 2, 4, 4)}, 10, None) == {}
    for case in S.decode_cases():
        maps = S.make_head_maps(case)
        want = odecode.generic_decode({k: v.clone() for k, v in maps.items()}, K=case['K'])
        out = {k: v.to(device) for k, v in maps.items()}
        got = generic_decode(out, case['K'], SimpleNamespace(zero_tracking=False))
        assert set(got.keys()) == set(want.keys()), case['name']
        for k, v in want.items():
            assert got[k].dtype == torch.float32 and tuple(got[k].shape) == tuple(v.shape), (case['name'], k)
            if k == 'kps_score':       # (mean over joints: summation order not restated)
                np.testing.assert_allclose(got[k].cpu().numpy(), v.numpy(), rtol=1e-5, atol=1e-7)
                continue
            np.testing.assert_array_equal(got[k].cpu().numpy(), v.numpy(), err_msg='%s.%s' % (case['name'], k))
    # zero_tracking mutates the caller's dict in place (decode.py:88-89)
    case = S.decode_cases()[0]
    maps = {k: v.to(device) for k, v in S.make_head_maps(case).items()}
    got = generic_decode(maps, case['K'], SimpleNamespace(zero_tracking=True))
    assert float(maps['tracking'].abs().max()) == 0.0 and float(got['tracking'].abs().max()) == 0.0
