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
        for b in range(case['B']):
            sc = ret['scores'][b].numpy()
            assert len(set(sc.tolist())) == len(sc), 'TEST DATA: %s image %d: two winners with exactly the same score' % (case['name'], b)
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


def _ref_detector(opt, sd):
    """the reference's Detector (detector.py) on CPU with ``sd`` instead of a checkpoint file"""
    import detector as ref_detector
    from model.model import create_model
    ref_detector.create_model = lambda arch, h, hc, opt=None: create_model(arch, h, hc, opt=opt)

    def fake_load(model, path, o):
        model.load_state_dict(sd)
        return model
    ref_detector.load_model = fake_load
    return ref_detector.Detector(opt)


def _prefetch_dict(images, meta):
    """the PrefetchDataset dict of test.py:31-48 collated with a leading batch dim of 1; pre_dets / cur_dets attached
    the way test.py:88-107 attaches them (plain lists, after the DataLoader)"""
    arrays = {k: v for k, v in meta.items() if k not in ('pre_dets', 'cur_dets')}
    pre = {'image': torch.zeros(1, 4, 4, 3), 'images': {1.0: images.unsqueeze(0)},
           'meta': {1.0: {k: torch.from_numpy(np.asarray(v)[None]) for k, v in arrays.items()}}}
    for k in ('pre_dets', 'cur_dets'):
        if k in meta:
            pre['meta'][k] = meta[k]
    return pre


def _calibrate_hm(case, opt, top=0.8, n_above=40):
    """per-class scale / bias of the hm output layer from the reference model's own logits on frame 0 (the pooled
    recipe of make_hm_calibration.py): every class standardised over the map, the pooled NMS peaks mapped so that the
    5th best scores ``top`` and the ``n_above``-th the threshold -- ~40 detections of mixed classes, scores apart"""
    import math
    import torch.nn.functional as F
    from model.model import create_model
    from scenarios import e2e_mode_frames
    sd = W.make_synthetic_state_dict(case['heads'], seed=case['seed'], hm_gain=1.0)
    model = create_model(opt.arch, opt.heads, opt.head_conv, opt=opt).eval()
    model.load_state_dict(sd)
    x = next(iter(e2e_mode_frames(case)))[0][:1]
    with torch.no_grad():
        raw = model(x, x, torch.zeros((1, 1, case['H'], case['W'])))[-1]['hm'][0].double()
    logit = lambda p: math.log(p / (1 - p))
    thr = case['opt']['track_thresh']
    mu, sg = raw.mean(dim=(1, 2)), raw.std(dim=(1, 2))
    z = (raw - mu.view(-1, 1, 1)) / sg.view(-1, 1, 1)
    pooled = torch.sort(z[F.max_pool2d(z[None], 3, 1, 1)[0] == z], descending=True)[0]
    lift = 1.0 if case['opt'].get('flip_test') else 0.0      # (the flip merge averages two maps: peaks flatten)
    G = (logit(top) - logit(thr)) / float(pooled[4] - pooled[n_above - 1])
    P = logit(top) + lift - G * float(pooled[4])
    scale = [float(G / sg[c]) for c in range(raw.shape[0])]
    bias = [float(P - scale[c] * (mu[c] + 4.6)) for c in range(raw.shape[0])]          # raw = w.h - 4.6 (prior bias)
    return {'scale': scale, 'bias': bias}


def gen_e2e_modes():
    """reference Detector.run in the modes scenarios.e2e_mode_cases() lists (T up to 16; Hungarian, max_age, public
    detections, flip_test, tracking,ddd with calib, 80 classes)"""
    from scenarios import e2e_mode_cases, e2e_mode_frames, e2e_mode_state_dict
    out, cal = {}, {}
    for case in e2e_mode_cases():
        opt = ref_opt(case['ref_args'] + ['--input_h', str(case['H']), '--input_w', str(case['W'])],
                      case['heads']['hm'], case['task'])
        assert dict(opt.heads) == dict(case['heads']), (opt.heads, case['heads'])
        if case['calibrated']:
            cal[case['name']] = _calibrate_hm(case, opt)
        calibration = cal.get(case['name'])
        if case.get('calibration_of'):                 # the full-size parity tests' weights (hm_calibration.json)
            with open(os.path.join(HERE, 'hm_calibration.json')) as f:
                calibration = json.load(f)[case['calibration_of']]
        det = _ref_detector(opt, e2e_mode_state_dict(case, calibration))
        frames = []
        for t, (images, meta) in enumerate(e2e_mode_frames(case)):
            ret = det.run(_prefetch_dict(images, meta))
            frames.append([{k: np.asarray(v, np.float64).tolist() for k, v in r.items()} for r in ret['results']])
            sc = [float(r['score']) for r in ret['results'] if int(r['age']) == 1]
            if case.get('calibration_of'):             # full size: a HAND-PICKED stream like the other full-size tests' --
                gaps = -np.diff(np.sort(np.array(sc))[::-1])        # no rank tie (< 5e-5) and no threshold tie (< 1e-4)
                edge = min(abs(s_ - th) for s_ in [float(r['score']) for r in ret['results']] for th in (0.4, 0.5))
                assert (len(gaps) == 0 or gaps.min() > 5e-5) and edge > 1e-4, ('TEST DATA: %s frame %d: rank gap %.1e / '
                                                                                'threshold distance %.1e: change the seed' % (case['name'], t, gaps.min(), edge))
            assert len(set(sc)) == len(sc), ('TEST DATA: %s frame %d holds two detections with exactly the same score (the '
                                             'order of exact ties in torch.topk is unspecified): change the seed' % (case['name'], t))
        out[case['name']] = frames
        ids = sorted({int(d['tracking_id']) for f in frames for d in f})
        print('%-14s dets/frame %s  ids %d  classes %s  carried %d' % (
            case['name'], [len(f) for f in frames], len(ids), sorted({int(d['class']) for f in frames for d in f}),
            sum(int(d['active']) == 0 for f in frames for d in f)))
    with open(os.path.join(HERE, 'e2e_modes_calibration.json'), 'w') as f:
        json.dump(cal, f)
    with open(os.path.join(HERE, 'e2e_modes.json'), 'w') as f:
        json.dump(out, f)
    print('e2e_modes.json', os.path.getsize(os.path.join(HERE, 'e2e_modes.json')))


def gen_model_full():
    """reference DLASeg forward at the BENCHMARKED sizes: 512 x 512 (BASELINE configs[1]) and 544 x 960 (the reference's
    own MOT input, datasets/mot.py:15: ragged 17 x 30 / 34 x 60 deep maps).  Stored: the full ``hm`` map and every
    regression head at stride 2 (a wrong halo / ragged-tile rule anywhere in the 50 layers shows in every pixel)"""
    from model.model import create_model
    out = {}
    heads = W.MOT_HEADS
    for name, h, w in (('mot_512', 512, 512), ('mot_544x960', 544, 960)):
        opt = ref_opt(['--pre_hm', '--ltrb_amodal', '--input_h', str(h), '--input_w', str(w)], heads['hm'])
        model = create_model(opt.arch, opt.heads, opt.head_conv, opt=opt).eval()
        model.load_state_dict(W.make_synthetic_state_dict(heads, seed=317))
        x, pre, hm = W.synthetic_inputs(1, h, w, seed=317)
        with torch.no_grad():
            y = model(x, pre, hm)[-1]
        for k, v in y.items():
            out['%s.%s' % (name, k)] = v.numpy() if k == 'hm' else v[:, :, ::2, ::2].contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, 'model_forward_full.npz'), **out)
    print('model_forward_full.npz', os.path.getsize(os.path.join(HERE, 'model_forward_full.npz')),
          {k: v.shape for k, v in out.items()})


def gen_opts():
    """the reference's own parsed option namespaces (opts().parse + update_dataset_info_and_set_heads, opts.py) for the
    experiments' command lines, as plain JSON: the GPU tests build the drop-in Detector from THESE (every attribute name the
    reference defines, none the drop-in invents) on a box where the reference does not exist"""
    out = {}
    for name, task, classes, args in (
            ('mot', 'tracking', 1, ['--pre_hm', '--ltrb_amodal', '--track_thresh', '0.4', '--pre_thresh', '0.5',
                                    '--input_h', '128', '--input_w', '160']),            # experiments/mot17_half.sh:5 (small input)
            ('kitti_flip', 'tracking', 3, ['--pre_hm', '--track_thresh', '0.4', '--flip_test', '--input_h', '128', '--input_w', '160']),
            ('nusc_ddd', 'tracking,ddd', 10, ['--pre_hm', '--track_thresh', '0.1', '--input_h', '128', '--input_w', '224']),
            ('mot_hungarian_public', 'tracking', 1, ['--pre_hm', '--ltrb_amodal', '--track_thresh', '0.4', '--pre_thresh', '0.5',
                                                     '--hungarian', '--public_det', '--load_results', 'x', '--max_age', '2',
                                                     '--input_h', '128', '--input_w', '160'])):
        o = ref_opt(args, classes, task)
        d = {}
        for k, v in sorted(vars(o).items()):
            if isinstance(v, (bool, int, float, str)) or v is None:
                d[k] = v
            elif isinstance(v, (list, tuple)) and all(isinstance(x, (bool, int, float, str)) for x in v):
                d[k] = list(v)
            elif isinstance(v, dict):
                d[k] = {kk: (list(vv) if isinstance(vv, (list, tuple)) else vv) for kk, vv in v.items()}
        out[name] = d
    with open(os.path.join(HERE, 'ref_opts.json'), 'w') as f:
        json.dump(out, f, indent=0)              # (no sort_keys: opt.heads keeps the reference's insertion order)
    print('ref_opts.json', {k: len(v) for k, v in out.items()})


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
    which = sys.argv[1:] or ['model', 'decode', 'post', 'tracker', 'prehm', 'e2e', 'writers', 'poseflip', 'e2emodes', 'modelfull', 'opts']
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
    if 'e2emodes' in which:
        gen_e2e_modes()
    if 'modelfull' in which:
        gen_model_full()
    if 'opts' in which:
        gen_opts()
