import json
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sd_crc(sd):
    c = 0
    for k in sd:
        c = zlib.crc32(np.ascontiguousarray(sd[k].numpy()).tobytes(), c)
    return c


def arr_crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def fixture_cfg(g):
    return json.loads(str(g['cfg']))


def require_same_rng(actual, expected, what):
    """Synthetic weights/inputs are regenerated from seeds; if this torch build
    draws different numbers the stored reference outputs do not apply."""
    if int(actual) != int(expected):
        pytest.skip('torch RNG stream differs from the fixture generator (%s)' % what)


@pytest.fixture(scope='session')
def hip_lib():
    from egonet_amd import _lib
    return _lib.lib()


def gpu_available():
    return torch.cuda.is_available()
