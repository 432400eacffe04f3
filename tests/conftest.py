import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running')


@pytest.fixture(scope='session')
def coco_skeleton0():
    import numpy as np
    from openpifpaf_amd import constants
    return np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
