#!/usr/bin/env python
"""Generate tests/golden/golden.npz: small input/output vectors of every stage of the hot path.

The reference (/root/reference, Iselix/kwiiyatta) cannot produce them: its arithmetic lives in
pyworld / pysptk / fastdtw / nnmnkwii, which are not installed and not installable here (SURVEY.md
8c).  The vectors are therefore outputs of oracle/ (the CPU restatement pinned by the reference's
known-answer tests, tests/test_oracle_kat.py) on a checked-in recording, and they pin
  * the oracle itself against drift (tests/test_golden.py, CPU), and
  * the HIP kernels on the GPU box, which has neither the reference nor needs the oracle for this.

    python tests/golden/make_golden.py          (run from the repository root; needs oracle/liboracle.so)
"""
import os
import sys

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as ko  # noqa: E402

WAV = os.path.join(ROOT, 'tests', 'data', 'cmu_us_clb_arctic', 'wav', 'arctic_a0001.wav')
FRAMES = np.array([0, 37, 58, 59, 60, 90, 121, 140])   # unvoiced, onset, voiced and gated frames


def main():
    fs, d = wavfile.read(WAV)
    x = np.ascontiguousarray(d[:int(0.75 * fs)].astype(np.float64) / 2 ** 15)   # first 0.75 s
    f0_dio, t = ko.dio(x, fs)
    f0 = ko.stonemask(x, f0_dio, t, fs)
    sp = ko.cheaptrick(x, f0, t, fs)
    ap = ko.d4c(x, f0, t, fs)
    y = ko.synthesize(f0, sp, ap, fs, 5.0)
    alpha = ko.mcepalpha(fs)
    mc = ko.sp2mc(sp, 24, alpha)
    sp_back = ko.mc2sp(mc[FRAMES], alpha, 2 * (sp.shape[1] - 1))

    # alignment: the utterance against a locally time-warped copy of itself
    rng = np.random.RandomState(7)
    idx = np.clip((np.arange(int(len(mc) * 1.15)) / 1.15 + 3 * np.sin(np.arange(int(len(mc) * 1.15)) / 9.0)), 0,
                  len(mc) - 1).astype(int)
    feat_x = np.ascontiguousarray(mc[:, 1:])
    feat_y = np.ascontiguousarray(mc[idx, 1:] + 0.01 * rng.standard_normal((len(idx), 24)))
    dist, path = ko.fastdtw(feat_x, feat_y, radius=4, dist=2)

    # conversion: a 2-component joint GMM with full covariances, static + delta + delta-delta (D = 2*72)
    d_static, D = 24, 72
    M = 2
    A = rng.standard_normal((M, 2 * D, 2 * D)) * 0.15
    covs = np.einsum('mij,mkj->mik', A, A) + 0.5 * np.eye(2 * D)
    means = rng.standard_normal((M, 2 * D)) * 0.3
    weights = np.array([0.4, 0.6])
    conv_in = np.ascontiguousarray(mc[40:100, 1:])
    conv = ko.gmm_mlpg(conv_in, weights, means, covs, diff=False)
    conv_diff = ko.gmm_mlpg(conv_in, weights, means, covs, diff=True)

    # MLSA differential filter: the first 0.4 s of the recording through a slowly varying filter
    T_ml, hop = 80, fs // 200
    walk = np.cumsum(rng.standard_normal((T_ml, 24)) * 0.02, axis=0) / np.arange(1, 25)
    mlsa_mc = np.hstack([np.zeros((T_ml, 1)), walk])
    mlsa_b = ko.mc2b(mlsa_mc, alpha)
    mlsa_x = np.ascontiguousarray(x[:T_ml * hop])
    mlsa_y = ko.mlsa_synthesis(mlsa_x, mlsa_b, alpha, hop)
    # aperiodicity codec
    ap_coded = ko.code_aperiodicity(np.ascontiguousarray(ap[FRAMES]), fs)
    ap_decoded = ko.decode_aperiodicity(ap_coded, fs, 512)

    out = os.path.join(HERE, 'golden.npz')
    np.savez_compressed(
        out, fs=fs, x=x, t=t, f0_dio=f0_dio, f0=f0, frames=FRAMES, sp_rows=sp[FRAMES], ap_rows=ap[FRAMES],
        sp_sum=sp.sum(axis=1), ap_mean=ap.mean(axis=1), y=y, alpha=alpha, mc=mc, sp_back=sp_back,
        feat_x=feat_x, feat_y=feat_y, dtw_dist=dist, dtw_path=np.asarray(path, dtype=np.int32),
        gmm_weights=weights, gmm_means=means, gmm_covs=covs, conv_in=conv_in, conv=conv, conv_diff=conv_diff,
        mlsa_mc=mlsa_mc, mlsa_b=mlsa_b, mlsa_y=mlsa_y, mlsa_hop=hop, ap_coded=ap_coded, ap_decoded=ap_decoded)
    print(out, os.path.getsize(out), 'bytes; frames', len(f0), 'path', len(path), 'dist', dist)


if __name__ == '__main__':
    main()
