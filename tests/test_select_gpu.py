"""The block-wide "sum of the m smallest of n" behind D4C's band aperiodicity (kwy_device.hpp) against a sort
on the host: spectra-like data, ties at the threshold, flat and degenerate inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, M = 2049, 1984        # D4C at 48 kHz: fft_size / 2 + 1 bins, all but the 65 largest


def _cases():
    rng = np.random.default_rng(11)
    k = np.arange(N)
    out = []
    # a main lobe over a decaying floor, as the windowed group-delay spectra look
    for width, floor in ((20.0, 1e-9), (6.0, 1e-14), (60.0, 1e-4)):
        c = rng.integers(100, N - 100)
        out.append(np.exp(-0.5 * ((k - c) / width) ** 2) * rng.uniform(0.5, 2.0) + floor * rng.random(N))
    out.append(rng.random(N))                                   # flat: the bin of the threshold holds hundreds of keys
    out.append(np.exp(rng.normal(0, 12, N)))                    # 100 dB of spread
    out.append(np.full(N, 3.25))                                # all equal
    out.append(np.zeros(N))                                     # all zero
    z = np.zeros(N); z[:40] = rng.random(40) + 1; out.append(z)  # fewer positive values than the cut: threshold is 0
    t = rng.random(N) * 1e-3; t[rng.choice(N, 90, replace=False)] = 7.0; out.append(t)   # 90 ties across the cut
    t = rng.random(N) * 1e-3; t[rng.choice(N, 65, replace=False)] = 7.0; out.append(t)   # ties end exactly at the cut
    s = np.exp(-k / 3.0); out.append(s)                         # the threshold lies 90 dB below the maximum
    d = rng.random(N) * 1e-300; d[5] = 1.0; out.append(d)       # overflow bin
    out.append(np.r_[np.full(70, 2.0 ** -1060), rng.random(N - 70) * 2.0 ** -1070])   # denormals
    return [np.ascontiguousarray(a, dtype=np.float64) for a in out]


@pytest.mark.parametrize('n,m', [(N, M), (N, N), (N, 1), (1025, 1025 - 33), (300, 290), (2304, 2000)])
def test_smallest_sum_matches_sort(n, m):
    import torch
    import ctypes
    import os
    from conftest import ROOT
    from kwiiyatta_amd import _lib                   # loads the HIP runtime the way the package does
    st = ctypes.CDLL(os.path.join(ROOT, 'kwiiyatta_amd', 'libkwy_selftest.so'))
    st.kwy_debug_smallest_sum_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p]
    cases = [c[:n] if len(c) >= n else np.resize(c, n) for c in _cases()]
    X = np.stack(cases)
    dev = torch.device('cuda', 0)
    dx = torch.from_numpy(X).to(dev)
    out = torch.zeros((len(cases), 2), dtype=torch.float64, device=dev)
    assert st.kwy_debug_smallest_sum_dev(torch.cuda.current_stream().cuda_stream, dx.data_ptr(), len(cases), n, m,
                                         out.data_ptr()) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, x in enumerate(cases):
        srt = np.sort(x)
        ref_small, ref_all = srt[:m].sum(), srt.sum()
        tol = 1e-13 * max(ref_all, 1e-300)
        assert abs(got[i, 1] - ref_all) <= tol, (i, got[i, 1], ref_all)
        assert abs(got[i, 0] - ref_small) <= tol, (i, got[i, 0], ref_small)
