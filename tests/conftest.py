import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def smpl_model_np():
    from soccerplayershapepose_amd import synthetic as S
    return S.smpl_model(seed=0)


@pytest.fixture(scope='session')
def regressor_sd_np():
    from soccerplayershapepose_amd import synthetic as S
    return S.regressor_state_dict(seed=0)
