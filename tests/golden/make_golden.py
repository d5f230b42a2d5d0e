#!/usr/bin/env python
"""Generate the committed golden vectors from the REFERENCE's own Python modules.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz|*.json.  Every fixture is the output of the reference's
own code (src/lib/model/networks/dla.py + base_model.py, model/decode.py,
model/utils.py, utils/post_process.py, utils/image.py, utils/tracker.py,
detector.py) on seeded synthetic inputs; tests/test_oracle_golden.py then checks
the oracle/ restatement against them (CPU, no reference needed), and the GPU
parity tests compare the HIP path with the oracle.  The DCNv2 op is the one piece
that is NOT the reference's code (un-vendored submodule): oracle/dcn_v2.DCN is
injected in its place, so DCN parity stays "unpinned" (see oracle/__init__.py).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from centertrack_amd import weights as W  # noqa: E402
from scenarios import (decode_cases, make_head_maps, postprocess_cases,  # noqa: E402
                       tracker_sequences, e2e_config, writer_case, pose_flip_inputs)

torch.set_num_threads(8)


def tonp(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def ref_opt(args, num_classes, task='tracking'):
    from opts import opts
    o = opts().parse([task, '--dataset', 'fake', '--load_model', 'x', '--gpus', '-1',
                      '--num_classes', str(num_classes)] + args)
    return opts().update_dataset_info_and_set_heads(o, ref_import.FakeDataset)


def gen_model():
    """reference DLASeg forward on seeded weights/inputs"""
    from model.model import create_model
    out = {}
    for name, heads, task, extra, (b, h, w) in [
            ('mot', W.MOT_HEADS, 'tracking', ['--pre_hm', '--ltrb_amodal'], (1, 64, 96)),
            ('nusc', W.NUSC_HEADS, 'tracking,ddd', ['--pre_hm'], (2, 64, 64))]:
        opt = ref_opt(extra + ['--input_h', str(h), '--input_w', str(w)], heads['hm'], task)
        assert dict(opt.heads) == dict(heads), (opt.heads, heads)
        model = create_model(opt.arch, opt.heads, opt.head_conv, opt=opt).eval()
        sd = W.make_synthetic_state_dict(heads, seed=317)
        assert set(model.state_dict().keys()) == set(sd.keys())
        model.load_state_dict(sd)
        x, pre, hm = W.synthetic_inputs(b, h, w, seed=317)
        with torch.no_grad():
            y = model(x, pre, hm)[-1]
            y1 = model(x, pre, None)[-1]     # pre_hm=None branch (dla.py:308-311)
        for k, v in y.items():
            out['%s.%s' % (name, k)] = v.numpy()
        out['%s_nohm.hm' % name] = y1['hm'].numpy()
        out['%s.wsum' % name] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
        out['%s.xsum' % name] = np.float64(float(x.double().abs().sum() + pre.double().abs().sum()
                                                 + hm.double().abs().sum()))
    np.savez_compressed(os.path.join(HERE, 'model_forward.npz'), **out)
    print('model_forward.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


def gen_decode():
    from model.decode import generic_decode
    out = {}
    for case in decode_cases():
        maps = make_head_maps(case)
        opt = type('O', (), {'zero_tracking': False})()
        inp = {k: v.clone() for k, v in maps.items()}
        with torch.no_grad():
            ret = generic_decode(inp, K=case['K'], opt=opt)
        for k, v in ret.items():
            out['%s.%s' % (case['name'], k)] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'decode.npz'), **out)
    print('decode.npz', len(out))


def gen_post_process():
    from utils.post_process import generic_post_process
    res = {}
    for case in postprocess_cases():
        opt = type('O', (), {'out_thresh': case['out_thresh']})()
        dets = {k: v.copy() for k, v in case['dets'].items()}
        r = generic_post_process(opt, dets, [case['c']], [case['s']], case['h'], case['w'],
                                 case['num_classes'], [case['calib']], case['height'], case['width'])
        items = []
        for it in r[0]:
            items.append({k: (np.asarray(v, dtype=np.float64).tolist()) for k, v in it.items()})
        res[case['name']] = items
    with open(os.path.join(HERE, 'post_process.json'), 'w') as f:
        json.dump(res, f)
    print('post_process.json', {k: len(v) for k, v in res.items()})


def gen_tracker():
    from utils.tracker import Tracker
    res = {}
    for seq in tracker_sequences():
        opt = type('O', (), seq['opt'])()
        tr = Tracker(opt)
        tr.init_track([dict(d) for d in seq.get('pre_dets', [])])
        frames = []
        for fr in seq['frames']:
            results = [{k: (np.array(v, np.float32) if isinstance(v, list) else v) for k, v in d.items()}
                       for d in fr['dets']]
            pub = fr.get('public_det')
            ret = tr.step(results, pub)
            frames.append([{'tracking_id': int(t['tracking_id']), 'age': int(t['age']),
                            'active': int(t['active']), 'score': float(t['score']),
                            'class': int(t['class'])} for t in ret])
        res[seq['name']] = frames
    with open(os.path.join(HERE, 'tracker.json'), 'w') as f:
        json.dump(res, f)
    print('tracker.json', {k: [len(f) for f in v] for k, v in res.items()})


def gen_e2e():
    """reference Detector.run over a short synthetic sequence (pre-processed-dict input path)"""
    import detector as ref_detector
    from model.model import create_model
    cfg = e2e_config()
    heads = cfg['heads']
    opt = ref_opt(cfg['ref_args'] + ['--input_h', str(cfg['H']), '--input_w', str(cfg['W'])],
                  heads['hm'])
    from scenarios import e2e_state_dict
    sd = e2e_state_dict(cfg)
    ref_detector.create_model = lambda arch, h, hc, opt=None: create_model(arch, h, hc, opt=opt)

    def fake_load(model, path, o):
        model.load_state_dict(sd)
        return model
    ref_detector.load_model = fake_load
    det = ref_detector.Detector(opt)
    from scenarios import e2e_frames
    out = {'pre_hm_sums': [], 'frames': []}
    for t, (images, meta) in enumerate(e2e_frames(cfg)):
        # the PrefetchDataset dict of test.py:31-48, collated with a leading batch dim of 1
        pre = {'image': torch.zeros(1, 4, 4, 3), 'images': {1.0: images.unsqueeze(0)},
               'meta': {1.0: {k: torch.from_numpy(np.asarray(v)[None]) for k, v in meta.items()}}}
        ret = det.run(pre)
        frame = []
        for r in ret['results']:
            frame.append({k: np.asarray(v, np.float64).tolist() for k, v in r.items()})
        out['frames'].append(frame)
    with open(os.path.join(HERE, 'e2e_mot.json'), 'w') as f:
        json.dump(out, f)
    print('e2e_mot.json', [len(f) for f in out['frames']],
          [[d['tracking_id'] for d in f] for f in out['frames']])


def gen_pre_hm():
    """reference Detector._get_additional_inputs + meta transforms"""
    import detector as ref_detector
    from scenarios import pre_hm_cases
    from utils.image import get_affine_transform
    out = {}
    for case in pre_hm_cases():
        d = ref_detector.Detector.__new__(ref_detector.Detector)
        d.opt = type('O', (), dict(pre_thresh=case['pre_thresh'], flip_test=case['flip_test'],
                                   device=torch.device('cpu')))()
        meta = case['meta']
        hm, inds = d._get_additional_inputs(case['tracks'], meta, with_hm=True)
        out[case['name'] + '.hm'] = hm.numpy()
        out[case['name'] + '.inds'] = inds.numpy()
        out[case['name'] + '.trans_input'] = get_affine_transform(
            meta['c'], meta['s'], 0, [meta['inp_width'], meta['inp_height']])
        out[case['name'] + '.trans_output_inv'] = get_affine_transform(
            meta['c'], meta['s'], 0, [meta['out_width'], meta['out_height']], inv=1)
    np.savez_compressed(os.path.join(HERE, 'pre_hm.npz'), **out)
    print('pre_hm.npz', {k: v.shape for k, v in out.items()})


def gen_pose_flip():
    """flip_lr / flip_lr_off of the reference (model/utils.py:33-50) on pose_flip_inputs(), COCO flip_idx"""
    from model.utils import flip_lr, flip_lr_off
    flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # datasets/coco_hp.py:18-19
    x = pose_flip_inputs()
    np.savez_compressed(os.path.join(HERE, 'pose_flip.npz'), hm_hp=flip_lr(x['hm_hp'].clone(), flip_idx).numpy(),
                        hps=flip_lr_off(x['hps'].clone(), flip_idx).numpy())
    print('pose_flip.npz')


def gen_writers():
    """MOT.save_results (datasets/mot.py:52-83) and KITTITracking.save_results (datasets/kitti_tracking.py:51-97) of
    the reference, run unmodified on writer_case(); pycocotools is stubbed (only imported, not used by the writers)"""
    import copy
    import tempfile
    import types
    for name in ('pycocotools', 'pycocotools.coco', 'pycocotools.cocoeval'):
        m = types.ModuleType(name)
        m.COCO = object
        m.COCOeval = object
        sys.modules[name] = m
    sys.modules['cv2'].__dict__.setdefault('imread', None)
    from dataset.datasets.mot import MOT
    from dataset.datasets.kitti_tracking import KITTITracking
    case = writer_case()
    out = {}
    for tag, cls, extra in (('mot', MOT, {'dataset_version': '17halfval'}),
                            ('kitti', KITTITracking, {'class_name': case['kitti_class_name']})):
        fake = types.SimpleNamespace(coco=types.SimpleNamespace(dataset={'videos': case['videos']}),
                                     video_to_images=case['video_to_images'], **extra)
        with tempfile.TemporaryDirectory() as d:
            cls.save_results(fake, copy.deepcopy(case['results']), d)
            files = {}
            for root, _, names in os.walk(d):
                for n in names:
                    with open(os.path.join(root, n)) as f:
                        files[os.path.relpath(os.path.join(root, n), d)] = f.read()
        out[tag] = files
    with open(os.path.join(HERE, 'writers.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('writers.json', {k: {n: len(t.splitlines()) for n, t in v.items()} for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['model', 'decode', 'post', 'tracker', 'prehm', 'e2e', 'writers', 'poseflip']
    if 'model' in which:
        gen_model()
    if 'decode' in which:
        gen_decode()
    if 'post' in which:
        gen_post_process()
    if 'tracker' in which:
        gen_tracker()
    if 'prehm' in which:
        gen_pre_hm()
    if 'e2e' in which:
        gen_e2e()
    if 'writers' in which:
        gen_writers()
    if 'poseflip' in which:
        gen_pose_flip()
