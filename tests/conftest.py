import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build them in-tree (hipcc cross-compiles
    # for gfx950 without a GPU).  Staleness is decided by content hashes (build.py), so this is a no-op -- a few
    # milliseconds of hashing -- when the library matches the sources, whatever the mtimes are
    from speaker_follower_amd import build
    build.build_lib(verbose=False)
    build.build_sim(verbose=False)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    return load
