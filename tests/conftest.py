import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, 'tests', 'data')
CLB_DIR = os.path.join(DATA, 'cmu_us_clb_arctic', 'wav')
SLT_DIR = os.path.join(DATA, 'cmu_us_slt_arctic', 'wav')
CLB_WAV = os.path.join(CLB_DIR, 'arctic_a0001.wav')
CLB_WAV2 = os.path.join(CLB_DIR, 'arctic_a0002.wav')
SLT_WAV = os.path.join(SLT_DIR, 'arctic_a0001.wav')


def clb_variant(suffix):
    return os.path.join(DATA, f'cmu_us_clb_arctic.{suffix}', 'wav', 'arctic_a0001.wav')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def round_equal(expect, actual, sig_dig=2):
    """The reference's check.round_equal (tests/conftest.py:46-51): `actual`
    truncated to `sig_dig` significant digits equals `expect`."""
    import math
    eps = math.pow(10, math.floor(math.log10(abs(expect)) - sig_dig + 1)) if expect != 0 else 0
    return expect <= actual < expect + eps


@pytest.fixture(scope='session')
def gpu_ctx():
    from kwiiyatta_amd import _lib
    return _lib.default_context()
