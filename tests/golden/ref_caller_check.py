#!/usr/bin/env python
"""The drop-in UNDER THE REFERENCE'S OWN CALLERS (build container only, CPU; own process: modules are stubbed).

    python tests/golden/ref_caller_check.py        -> one JSON line

B1  ``src/lib/model/networks/dla.py:19`` does ``from .DCNv2.dcn_v2 import DCN``.  Here that module IS the two-line shim
    of INTEGRATION.md section A (``from centertrack_amd.dcn_v2 import DCN, DCNv2, dcn_v2_conv``); the reference's
    ``create_model('dla_34', heads, 256, opt)`` (model/model.py:24-33 -> DLASeg, dla.py:577-640) then builds its 16
    ``DeformConv`` nodes (dla.py:506-518) on ``centertrack_amd.dcn_v2.DCN``.  Checked: every node's conv is that class;
    the state dict of the reference model has EXACTLY the keys and shapes of ``DLASegHIP`` (B2), for the MOT and the
    nuScenes head sets.
B2  checkpoints cross both ways: the reference model's state dict, saved in the reference's checkpoint format
    (``{'epoch', 'state_dict'}``, model/model.py:92-104), is loaded by ``centertrack_amd.model.load_model`` into
    ``DLASegHIP``; a ``DLASegHIP`` checkpoint is loaded by the REFERENCE's ``load_model`` (model.py:35-90) into the
    reference model -- tensors identical afterwards, no "missing / unexpected key" path taken.
B4  ``test.py``'s ``PrefetchDataset`` (test.py:21-52) is instantiated from the reference's own file with
    ``centertrack_amd.detector.Detector.pre_process`` as its ``pre_process_func`` (test.py:75) over a fake dataset of
    synthetic frames (cv2.imread stubbed), iterated through a ``torch.utils.data.DataLoader`` like test.py:74-76, and
    the collated dict is unpacked by ``Detector.parse_prefetched`` -- the code ``Detector.run`` uses for it
    (detector.py:84-92): images / meta equal ``pre_process`` called directly; ``is_first_frame`` / ``video_id`` are there
    for the caller's ``reset_tracking`` (test.py:88-98); with --flip_test the batch is 2.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

REF = ref_import.install()
import torch  # noqa: E402

# ---- B1: the INTEGRATION.md shim in place of the un-vendored submodule ----
SHIM = 'from centertrack_amd.dcn_v2 import DCN, DCNv2, dcn_v2_conv  # noqa: F401  (HIP, MI355X)\n'
shim = types.ModuleType('model.networks.DCNv2.dcn_v2')
exec(compile(SHIM, 'src/lib/model/networks/DCNv2/dcn_v2.py', 'exec'), shim.__dict__)
sys.modules['model.networks.DCNv2.dcn_v2'] = shim
sys.modules['model.networks.DCNv2'].dcn_v2 = shim

from centertrack_amd import dcn_v2 as hip_dcn  # noqa: E402
from centertrack_amd import model as hip_model  # noqa: E402
from centertrack_amd import weights as W  # noqa: E402
from centertrack_amd.detector import Detector  # noqa: E402


def ref_opt(args, num_classes, task='tracking'):
    from opts import opts
    o = opts().parse([task, '--dataset', 'fake', '--load_model', 'x', '--gpus', '-1',
                      '--num_classes', str(num_classes)] + args)
    return opts().update_dataset_info_and_set_heads(o, ref_import.FakeDataset)


def check_models(out):
    from model.model import create_model, load_model, save_model
    from model.networks import dla as ref_dla
    assert ref_dla.DCN is hip_dcn.DCN, 'dla.py did not import the shim'
    for name, heads, task, extra in (('mot', W.MOT_HEADS, 'tracking', ['--pre_hm', '--ltrb_amodal']),
                                     ('nusc', W.NUSC_HEADS, 'tracking,ddd', ['--pre_hm'])):
        opt = ref_opt(extra, heads['hm'], task)
        assert dict(opt.heads) == dict(heads)
        ref = create_model(opt.arch, opt.heads, opt.head_conv, opt=opt)
        nodes = [m for m in ref.modules() if isinstance(m, ref_dla.DeformConv)]
        assert len(nodes) == 16 and all(type(m.conv) is hip_dcn.DCN for m in nodes)
        ours = hip_model.create_model(opt.arch, opt.heads, opt.head_conv, opt=opt)
        rsd, osd = ref.state_dict(), ours.state_dict()
        assert set(rsd.keys()) == set(osd.keys()) and len(rsd) == len(osd), 'state-dict keys differ'    # (order: heads first in the reference, last here; loading is by name)
        assert all(tuple(rsd[k].shape) == tuple(osd[k].shape) and rsd[k].dtype == osd[k].dtype for k in rsd)
        sd = W.make_synthetic_state_dict(heads, seed=317)
        with tempfile.TemporaryDirectory() as d:
            # reference -> ours
            ref.load_state_dict(sd)
            p1 = os.path.join(d, 'ref.pth')
            save_model(p1, 7, ref)                                            # model/model.py:92-104
            o2 = types.SimpleNamespace(resume=False, reuse_hm=False, reset_hm=False, lr=1e-4, lr_step=[])
            ours = hip_model.load_model(ours, p1, o2)
            osd = ours.state_dict()
            assert all(torch.equal(osd[k], sd[k]) for k in sd)
            # ours -> reference (the reference's own loader, its shape / missing-key reporting untouched)
            sd2 = W.make_synthetic_state_dict(heads, seed=5)
            ours.load_state_dict(sd2)
            p2 = os.path.join(d, 'ours.pth')
            torch.save({'epoch': 3, 'state_dict': ours.state_dict()}, p2)
            ref = load_model(ref, p2, o2)
            rsd = ref.state_dict()
            assert all(torch.equal(rsd[k], sd2[k]) for k in sd2)
        out[name] = {'keys': len(rsd), 'deform_nodes': len(nodes), 'params': int(sum(v.numel() for v in rsd.values()))}


def check_prefetch(out):
    sys.path.insert(0, os.path.join(REF, 'src'))
    sys.modules.setdefault('_init_paths', types.ModuleType('_init_paths'))
    logger = types.ModuleType('logger')
    logger.Logger = object
    sys.modules['logger'] = logger
    import cv2
    frames = {'%06d.jpg' % i: np.random.RandomState(i).randint(0, 256, (375, 1242, 3)).astype(np.uint8) for i in range(3)}
    cv2.imread = lambda path: frames[os.path.basename(path)].copy()
    import test as ref_test                                                     # the reference's src/test.py
    for flip in (False, True):
        opt = ref_opt(['--pre_hm', '--input_h', '384', '--input_w', '1280'] + (['--flip_test'] if flip else []), 3)
        ref_test.opt = opt                                                      # (test.py:35 reads the module global)
        infos = {i + 1: {'file_name': '%06d.jpg' % i, 'frame_id': i + 1, 'video_id': 4,
                         'calib': np.arange(12, dtype=np.float32).reshape(3, 4)} for i in range(3)}
        dataset = types.SimpleNamespace(
            images=[1, 2, 3], img_dir='/nowhere', coco=types.SimpleNamespace(loadImgs=lambda ids: [infos[i] for i in ids]),
            get_default_calib=lambda w, h: np.zeros((3, 4), np.float32))
        det = Detector.__new__(Detector)
        det._init_host(opt)                                                     # no device: what a worker process touches
        loader = torch.utils.data.DataLoader(ref_test.PrefetchDataset(opt, dataset, det.pre_process),
                                             batch_size=1, shuffle=False, num_workers=0, pin_memory=False)
        seen = 0
        for ind, (img_id, pre) in enumerate(loader):
            images, meta = det.parse_prefetched(pre)
            want_images, want_meta = det.pre_process(frames[infos[ind + 1]['file_name']], 1.0,
                                                     {'calib': infos[ind + 1]['calib']})
            assert images.shape == ((2 if flip else 1), 3, 384, 1280) and images.dtype == torch.float32
            assert torch.equal(images, want_images)
            assert set(meta) == set(want_meta), (sorted(meta), sorted(want_meta))
            for k, v in want_meta.items():
                np.testing.assert_array_equal(np.asarray(meta[k]), np.asarray(v), err_msg=k)
            assert ('is_first_frame' in pre) == (ind == 0)
            if ind == 0:
                assert int(pre['video_id']) == 4
            assert int(img_id) == ind + 1 and tuple(pre['image'].shape) == (1, 375, 1242, 3)
            seen += 1
        assert seen == 3
        out['prefetch_flip%d' % int(flip)] = {'frames': seen, 'meta_keys': sorted(want_meta)}
    # one test scale only: refused loudly, not silently truncated (detector.py:78)
    opt = ref_opt(['--test_scales', '1,1.5'], 1)
    try:
        Detector.__new__(Detector)._init_host(opt)
        out['multi_scale_refused'] = False
    except Exception as e:
        out['multi_scale_refused'] = 'one test scale' in str(e)


if __name__ == '__main__':
    out = {}
    check_models(out)
    check_prefetch(out)
    print(json.dumps(out))
