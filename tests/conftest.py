import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """a fresh checkout carries no built artefacts (they are git-ignored): build the HIP library and the oracle's C
    restatement once, exactly as __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU)"""
    # the CPU oracle (torch conv / gather on the host) is fastest at ~16 threads: on the GPU box's 256 hardware threads
    # the default (one thread per core) makes every oracle frame several times slower (bench.py's thread sweep)
    try:
        import torch
        if (os.cpu_count() or 1) > 16:
            torch.set_num_threads(16)
    except Exception:
        pass
    from centertrack_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
