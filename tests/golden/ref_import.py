"""Import the reference's own Python modules (build container only).

Used ONLY by tests/golden/make_golden.py to generate the committed golden vectors;
/root/reference does not exist on the GPU box and nothing in tests/, smoke() or
bench.py imports this at run time.  Third-party modules absent from the image are
stubbed:

  torchvision                       imported by model/model.py:5, never used
  cv2                               getAffineTransform -> exact 3-point solve
  numba, progress                   imported, unused on this path
  sklearn.utils.linear_assignment_  scipy linear_sum_assignment
  model.networks.DCNv2.dcn_v2       the un-vendored DCNv2 submodule: oracle/dcn_v2.DCN
                                    (=> DCNv2 parity is UNPINNED, everything else is
                                    the reference's own code)
  dataset.dataset_factory           needs pycocotools; only class attributes are read
  utils.debugger                    cv2/matplotlib drawing, not on the timed path
"""
import os
import sys
import types

import numpy as np

# /root/reference is read-only for this project: importing its modules must not leave __pycache__ directories in it
# (this process and every child process it starts)
sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

REF = os.environ.get('CENTERTRACK_REFERENCE', '/root/reference')
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))


class FakeDataset(object):
    """class attributes the hot path reads (detector.py:39-47; datasets/*.py)"""
    num_categories = 1
    default_resolution = [544, 960]
    mean = np.array([0.40789654, 0.44719302, 0.47026115], dtype=np.float32).reshape(1, 1, 3)
    std = np.array([0.28863828, 0.27408164, 0.27809835], dtype=np.float32).reshape(1, 1, 3)
    rest_focal_length = 1200
    flip_idx = []
    num_joints = 17


def install():
    if not os.path.isdir(REF):
        raise RuntimeError('reference checkout not found at %s' % REF)
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    lib = os.path.join(REF, 'src', 'lib')
    if lib not in sys.path:
        sys.path.insert(0, lib)
    from oracle.image import get_affine_transform_3pt
    from oracle import dcn_v2 as oracle_dcn
    from oracle.tracker import hungarian_assignment as linear_assignment

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = mod('torchvision')
    tv.models = mod('torchvision.models')
    tv.models.utils = mod('torchvision.models.utils', load_state_dict_from_url=lambda *a, **k: {})
    mod('cv2', getAffineTransform=get_affine_transform_3pt, INTER_LINEAR=1)
    mod('numba', jit=lambda *a, **k: (lambda f: f))
    pr = mod('progress')
    pr.bar = mod('progress.bar', Bar=object)
    import sklearn.utils  # noqa: F401
    mod('sklearn.utils.linear_assignment_', linear_assignment=linear_assignment)
    pkg = mod('model.networks.DCNv2')
    pkg.__path__ = []
    pkg.dcn_v2 = mod('model.networks.DCNv2.dcn_v2', DCN=oracle_dcn.DCN)
    mod('dataset.dataset_factory', get_dataset=lambda name: FakeDataset,
        dataset_factory={'fake': FakeDataset})

    class Debugger(object):
        def __init__(self, *a, **k):
            self.imgs = {}

        def clear(self):
            pass

    mod('utils.debugger', Debugger=Debugger)
    import torch
    torch.cuda.synchronize = lambda *a, **k: None   # detector.py:139,338,344,348 (CPU host)
    return REF
