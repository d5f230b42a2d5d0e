"""create_model / load_model (reference src/lib/model/model.py:24-90, inference part) on CPU: checkpoint layout,
DataParallel prefix, dropped / missing / mis-shaped parameters, --reset_hm / --reuse_hm."""
import types

import torch

from centertrack_amd import weights as W
from centertrack_amd.model import DLASegHIP, create_model, load_model


def _ckpt(tmp_path, heads, prefix='module.', extra=True):
    sd = W.make_synthetic_state_dict(heads, seed=5)
    out = {prefix + k: v for k, v in sd.items()}
    if extra:
        out[prefix + 'not_a_parameter.weight'] = torch.zeros(3)
    p = str(tmp_path / 'model.pth')
    torch.save({'epoch': 70, 'state_dict': out}, p)
    return p, sd


def test_checkpoint_round_trip_with_dataparallel_prefix(tmp_path, capsys):
    p, sd = _ckpt(tmp_path, W.MOT_HEADS)
    opt = types.SimpleNamespace(reset_hm=False, reuse_hm=False)
    model = load_model(create_model('dla_34', W.MOT_HEADS, 256, opt=None), p, opt)
    got = model.state_dict()
    assert set(got.keys()) == set(sd.keys())
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    out = capsys.readouterr().out
    assert 'epoch 70' in out and 'Drop parameter not_a_parameter.weight.' in out


def test_coco_heatmap_head_reused_or_reset_for_another_class_count(tmp_path, capsys):
    p, sd = _ckpt(tmp_path, W.COCO_HEADS, prefix='', extra=False)          # 80-class checkpoint
    m_skip = load_model(DLASegHIP(W.KITTI_HEADS), p, types.SimpleNamespace(reset_hm=False, reuse_hm=False))
    assert 'Skip loading parameter hm.2.weight' in capsys.readouterr().out
    assert tuple(m_skip.state_dict()['hm.2.weight'].shape) == (3, 256, 1, 1)
    assert torch.equal(m_skip.state_dict()['wh.2.weight'], sd['wh.2.weight'])
    m_reuse = load_model(DLASegHIP(W.KITTI_HEADS), p, types.SimpleNamespace(reset_hm=False, reuse_hm=True))
    assert 'Reusing parameter hm.2.weight' in capsys.readouterr().out
    assert torch.equal(m_reuse.state_dict()['hm.2.weight'], sd['hm.2.weight'][:3])
    assert torch.equal(m_reuse.state_dict()['hm.2.bias'], sd['hm.2.bias'][:3])
    assert torch.equal(m_reuse.state_dict()['hm.0.weight'], sd['hm.0.weight'])  # same shape: loaded as is
    # --reset_hm: an 80-row hm parameter is not loaded even where the shapes agree
    m_reset = load_model(DLASegHIP(W.COCO_HEADS), p, types.SimpleNamespace(reset_hm=True, reuse_hm=False))
    assert not torch.equal(m_reset.state_dict()['hm.2.weight'], sd['hm.2.weight'])
    assert torch.equal(m_reset.state_dict()['reg.2.weight'], sd['reg.2.weight'])
