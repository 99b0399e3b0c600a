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
        got = rs.abs_normal(EPS / 48000, (100, 1025))
        rs.sync()
        got = got.cpu().numpy()
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
        got = rs.normal(0.25, 3.0, (n,))
        rs.sync()
        got = got.cpu().numpy()
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
    a_dev = rs.abs_normal(1e-20, (100, 513))
    rs.to_global()                    # (synchronises the generator's stream)
    a_dev = a_dev.cpu().numpy()
    assert np.abs(a_dev / a - 1).max() <= 1e-15
    assert np.allclose(np.random.normal(size=5), b, rtol=1e-15, atol=0)


def test_pair_blocks_in_one_call_and_both_code_paths():
    """the four pad blocks of a pair in one call (the five-launch path: serial MT19937 words, chip-wide accept / scan /
    emit) equal four separate numpy calls, the single-workgroup kernel (small requests) continues the same stream, and
    the states agree after every mix of the two"""
    import torch
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    ref = np.random.RandomState(5)
    rs = DeviceRandomState.from_seed(5)
    for shape in ((100, 1025), (100, 513), (3, 7), (100, 1025), (41, 101)):
        outs = [torch.empty(shape, dtype=torch.float64, device='cuda') for _ in range(4)]
        rs.abs_normal_blocks(EPS / 48000, outs)
        rs.sync()
        for t in outs:
            want = np.abs(ref.normal(0, EPS / 48000, shape))
            assert np.abs(t.cpu().numpy() / want - 1).max() <= 1e-15
        assert _same_state(rs.get_state(), ref.get_state())
        odd = rs.normal(1.0, 2.0, (4099,))        # large and odd: the cached twin crosses into the next call
        rs.sync()
        assert np.abs(odd.cpu().numpy() - ref.normal(1.0, 2.0, 4099)).max() <= 1e-14
        assert _same_state(rs.get_state(), ref.get_state())
