import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture
def testing_lib():
    """Routes the test through libtdship_testing.so (the product's sources + the header's "testing hooks": forced strip widths, the
    ring-walk fallback of K2b, ablation switches) and yields it.  Handles created by the test die with its locals, before the switch back."""
    from torchdrivesim_amd import _native
    with _native.testing() as T:
        yield T
