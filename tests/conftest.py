import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'oracle'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def model_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'model_goldens.npz'))


@pytest.fixture(scope='session')
def postproc_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'postproc_goldens.npz'), allow_pickle=True)


@pytest.fixture(scope='session')
def head_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'head_goldens.npz'))


@pytest.fixture(scope='session')
def label_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'label_goldens.npz'))


@pytest.fixture(scope='session')
def qe_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'qe_goldens.npz'))
