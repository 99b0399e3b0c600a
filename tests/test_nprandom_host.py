"""The host half of the generator's jump-ahead (kwy_nprandom.hip): MT19937's characteristic polynomial by
Berlekamp-Massey, x^(624 * 512 * s - 1) mod phi for the segments, and the library's own self-check -- no GPU needed.
The polynomial of a segment is checked here against a plain Python restatement of the recurrence seeded like numpy:
the key of segment s is the XOR of the head's windows the polynomial selects
(reference: np.random.normal behind /root/reference/kwiiyatta/vocoder/world.py:158-161)."""
import ctypes

import numpy as np


def _stream(key, nwords):
    z = [int(v) for v in key]
    for t in range(624, nwords):
        y = (z[t - 624] & 0x80000000) | (z[t - 623] & 0x7fffffff)
        z.append(z[t - 624 + 397] ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0))
    return z


def test_jump_polynomials_reproduce_the_stream():
    from kwiiyatta_amd._lib import lib
    f = lib.kwy_np_jump_tables_host
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_void_p]
    nseg = 2
    words = np.zeros((nseg, 624), dtype=np.uint32)
    assert f(nseg, words.ctypes.data) == 0            # includes the library's self-check
    key = np.random.RandomState(31337).get_state()[1]
    seg_words = 624 * 512
    z = _stream(key, seg_words * nseg + 624)
    head = np.array(z[:624 * 33], dtype=np.uint32)
    for s in range(1, nseg + 1):
        bits = np.unpackbits(words[s - 1].view(np.uint8), bitorder='little')
        taps = np.nonzero(bits)[0]
        assert taps.max() < 19937 and 5000 < len(taps) < 15000      # a dense polynomial of degree < 19937
        for k in (0, 1, 227, 396, 623):
            assert int(np.bitwise_xor.reduce(head[k + 1 + taps])) == z[seg_words * s + k], (s, k)
    assert f(0, None) != 0 and f(5000, None) != 0     # out of range: refused
