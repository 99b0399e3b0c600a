"""Feature-distance metrics used by the reference's known-answer tests
(semantics of /root/reference/tests/feature.py:13-60, restated):
relative mean-frame-norm difference, after trimming to the shorter length."""
import math

import numpy as np
import scipy.signal as ssig


def _reshape_axis(feature, new_len):
    old = feature.shape[1]
    g = math.gcd(old, new_len)
    pad = old // g * 20
    trim = new_len // g * 20
    padded = np.hstack((np.repeat(feature[:, :1], pad, axis=1), feature,
                        np.repeat(feature[:, -1:], pad, axis=1)))
    return ssig.resample_poly(padded, new_len, old, axis=1)[:, trim:-trim]


def calc_diff(expected, actual, strict=True):
    assert not strict or abs(len(expected) - len(actual)) <= 1
    n = min(len(expected), len(actual))
    expected, actual = expected[:n], actual[:n]
    if expected.ndim > 1:
        if expected.shape[1] < actual.shape[1]:
            expected = _reshape_axis(expected, actual.shape[1])
        elif actual.shape[1] < expected.shape[1]:
            actual = _reshape_axis(actual, expected.shape[1])
        norm = lambda v: np.linalg.norm(v, axis=1)  # noqa: E731
    else:
        norm = np.abs
    return np.mean(norm(expected - actual)) / np.mean(norm(expected))


def calc_powered_diff(expected, actual, **kw):
    return calc_diff(np.sqrt(expected), np.sqrt(actual), **kw)


def feature_diffs(exp, act, **kw):
    """exp/act: dicts with f0, sp, ap, mc -> (f0, spec, ap, mcep) diffs."""
    return (calc_diff(exp['f0'], act['f0'], **kw),
            calc_powered_diff(exp['sp'], act['sp'], **kw),
            calc_diff(exp['ap'], act['ap'], **kw),
            calc_diff(exp['mc'], act['mc'], **kw))
