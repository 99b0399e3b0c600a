import sys, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from kwiiyatta_amd.backend import dtw
rng = np.random.default_rng(0)
def series(T, dim, warp):
    t = np.linspace(0, 1, T) ** warp
    base = np.stack([np.sin(2 * np.pi * (k + 1) * t * 3 + k) for k in range(dim)], 1)
    return base + 0.05 * rng.standard_normal((T, dim))
x = series(2201, 26, 1.0); y = series(2401, 26, 1.3)
for _ in range(2):
    d, p = dtw.fastdtw(x, y, radius=32)
print(d, len(p))
