"""kwy_finish_pcm16_batch_dev reproduces numpy's float64 mean to the last bit (csrc/kwy_finish.hip: chunks of
np.getbufsize() elements added in order, each chunk by numpy's pairwise routine).  That model of numpy is an
observation, not a documented contract: this CPU test restates it in Python and holds it against the numpy that is
installed, so that a numpy whose reduction works differently fails HERE, by name, and not as an off-by-one-LSB sample
on the GPU box.  (The reference's post-step: kwiiyatta/wavfile.py:17-21, vocoder/abc/synthesizer.py:14.)"""
import numpy as np

CHUNK, LEAF = 8192, 128


def _leaf(a):
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r += v
        return r
    r = [float(a[j]) for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += a[i]
        i += 1
    return res


def _pairwise(a):
    n = len(a)
    if n <= LEAF:
        return _leaf(a)
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(a[:n2]) + _pairwise(a[n2:])


def model_sum(a):
    s = 0.0
    for i in range(0, len(a), CHUNK):
        s += _pairwise(a[i:i + CHUNK])
    return s


def test_buffer_size_is_the_chunk_the_kernel_assumes():
    assert np.getbufsize() == CHUNK


def test_numpy_sum_is_chunked_pairwise():
    rng = np.random.default_rng(11)
    for n in (1, 7, 8, 9, 127, 128, 129, 143, 255, 1000, 8191, 8192, 8193, 8192 + 135, 3 * 8192 + 4103, 52801, 100003):
        for _ in range(2):
            a = rng.standard_normal(n) * rng.uniform(0.05, 5.0) + rng.uniform(-1.0, 1.0)
            lst = a.tolist()
            assert model_sum(lst) == float(np.add.reduce(a)), n
            assert model_sum(lst) / n == float(a.mean()), n
            b = a[3:]                                   # a view that starts off the allocation's alignment
            assert model_sum(b.tolist()) == float(b.sum()), n


def test_node_below_depth_six_is_at_most_two_leaves():
    """the kernel lets one lane evaluate a depth-6 node as one leaf or two: a chunk's nodes at that depth never exceed
    128 + 15 elements"""
    def depth6(n, d=0):
        if n <= LEAF or d == 6:
            return [n]
        n2 = n // 2
        n2 -= n2 % 8
        return depth6(n2, d + 1) + depth6(n - n2, d + 1)
    for L in list(range(1, 600)) + list(range(8192 - 600, 8193)) + [4096, 4097, 6000, 7777]:
        sizes = depth6(L)
        assert len(sizes) <= 64 and max(sizes) <= LEAF + 15, L
        for m in sizes:
            if m > LEAF:
                h = m // 2
                h -= h % 8
                assert h <= LEAF and m - h <= LEAF
