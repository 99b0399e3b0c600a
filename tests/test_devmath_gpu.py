"""kwy_log / kwy_sincos_medium (kwy_device.hpp: the fdlibm forms the per-bin loops of CheapTrick and of the synthesis
pulses call instead of the library's log / sincos) against numpy, in ulps."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_log_and_sincos_against_numpy():
    import torch
    from conftest import ROOT
    from kwiiyatta_amd import _lib                   # noqa: F401  (loads the HIP runtime the way the package does)
    st = ctypes.CDLL(os.path.join(ROOT, 'kwiiyatta_amd', 'libkwy_selftest.so'))
    st.kwy_debug_devmath_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    rng = np.random.default_rng(3)
    pos = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2.0, 200000),
                          1.0 + rng.uniform(-1e-6, 1e-6, 1000), [1.0, 0.5, 2.0, 2.0 ** -1022, 5e-324, 3e-310, 1.7e308]])
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-300, 300, 200000),
                          rng.uniform(-1e6, 1e6, len(pos) - 400000)])
    ang[:9] = [0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, np.pi / 4, 3 * np.pi / 4, 1e-300, -1e-9]
    assert len(ang) == len(pos)
    dev = torch.device('cuda', 0)
    outs = [torch.empty(len(pos), dtype=torch.float64, device=dev) for _ in range(6)]
    for x, (lg, sn, cs) in ((pos, outs[:3]), (ang, outs[3:])):
        dx = torch.from_numpy(x).to(dev)
        assert st.kwy_debug_devmath_dev(torch.cuda.current_stream().cuda_stream, dx.data_ptr(), len(x), lg.data_ptr(),
                                        sn.data_ptr(), cs.data_ptr()) == 0
        torch.cuda.synchronize()
    lg = outs[0].cpu().numpy()
    sn, cs = outs[4].cpu().numpy(), outs[5].cpu().numpy()
    ref = np.log(pos)
    nz = ref != 0
    assert _ulps(lg[nz], ref[nz]).max() <= 1.0
    assert (lg[~nz] == 0).all()
    # sin / cos: 1.5 ulp of the result (+ the 1e-33 |k| the two-constant reduction leaves of pi/2)
    for got, ref in ((sn, np.sin(ang)), (cs, np.cos(ang))):
        assert (np.abs(got - ref) <= 1.5 * np.spacing(np.abs(ref)) + 1e-27).all()
    # the special values of log
    sp = np.array([0.0, -1.0, np.inf, np.nan])
    d = torch.from_numpy(sp).to(dev)
    o = [torch.empty(4, dtype=torch.float64, device=dev) for _ in range(3)]
    assert st.kwy_debug_devmath_dev(torch.cuda.current_stream().cuda_stream, d.data_ptr(), 4, o[0].data_ptr(),
                                    o[1].data_ptr(), o[2].data_ptr()) == 0
    torch.cuda.synchronize()
    r = o[0].cpu().numpy()
    assert r[0] == -np.inf and np.isnan(r[1]) and r[2] == np.inf and np.isnan(r[3])
