"""numpy's legacy normal generator on the GPU (kwy_nprandom.hip) against numpy itself: same accept / reject pattern
(so the same number of MT19937 words consumed and the same state afterwards), values within a few ulp (the device's
log() may round differently from glibc's), the cached second Gaussian handled like numpy does across calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = 2.220446049250313e-16


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and \
        (a[4] == b[4] or abs(a[4] - b[4]) <= 4e-16 * abs(b[4]))


@pytest.mark.parametrize('seed', [0, 1, 1234, 2 ** 31 - 1])
def test_pad_block_equals_numpy(seed):
    """the reference's call: np.abs(np.random.normal(0, EPS / fs, (100, 1025))), four blocks in a row"""
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    rs = DeviceRandomState.from_seed(seed)
    ref = np.random.RandomState(seed)
    for block in range(4):
        got = rs.abs_normal(EPS / 48000, (100, 1025)).cpu().numpy()
        want = np.abs(ref.normal(0, EPS / 48000, (100, 1025)))
        assert got.shape == want.shape
        assert np.abs(got / want - 1).max() <= 1e-15, block
        assert (got == want).mean() > 0.9
    assert _same_state(rs.get_state(), ref.get_state())


def test_odd_counts_and_cached_gaussian():
    """odd request sizes leave the twin value cached (has_gauss) for the next call; tiny and empty requests; a start in
    the middle of an MT19937 block after other draws moved the position"""
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    ref = np.random.RandomState(99)
    ref.random_sample(5)              # 10 words: the position is not a multiple of four any more
    ref.randint(0, 10, 3)             # and a few single words
    rs = DeviceRandomState(ref.get_state())
    for n in (1, 1, 2, 3, 0, 7, 155, 156, 157, 311, 312, 313, 1000, 1, 4097, 12345):
        got = rs.normal(0.25, 3.0, (n,)).cpu().numpy()
        want = ref.normal(0.25, 3.0, n)
        assert got.shape == want.shape
        if n:
            assert np.abs(got - want).max() <= 1e-14, n
        assert _same_state(rs.get_state(), ref.get_state()), n
    # continue on the host from the device's state and the other way round
    host = np.random.RandomState()
    host.set_state(rs.get_state())
    assert np.array_equal(host.normal(size=10), ref.normal(size=10))


def test_global_generator_round_trip():
    """np.random.seed -> device draws -> np.random continues exactly where numpy alone would be"""
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    np.random.seed(7)
    a = np.abs(np.random.normal(0, 1e-20, (100, 513)))
    b = np.random.normal(size=5)
    np.random.seed(7)
    rs = DeviceRandomState.from_global()
    a_dev = rs.abs_normal(1e-20, (100, 513)).cpu().numpy()
    rs.to_global()
    assert np.abs(a_dev / a - 1).max() <= 1e-15
    assert np.allclose(np.random.normal(size=5), b, rtol=1e-15, atol=0)
