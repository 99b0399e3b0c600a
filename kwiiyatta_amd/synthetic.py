"""Deterministic synthetic utterances for the benchmark and the full-size tests.

Generator of SURVEY.md section 8(d), config 2/3: a voiced/unvoiced schedule
(0.40 s voiced, 0.10 s unvoiced => 80 % voiced), f0(t) = base * 2^(0.4 sin(2 pi 0.7 t)),
a harmonic sum up to Nyquist shaped by a slowly moving 4-formant envelope, plus
white noise at -35 dB (voiced) / -20 dB (unvoiced); peak 0.6.  The f0 track
handed to the hot path is the generating contour sampled every frame_period
(0 where unvoiced), so DIO/StoneMask stay outside the timed region.
"""
import numpy as np


def make_utterance(seed=1234, fs=48000, seconds=10.0, frame_period=5.0, f0_base=140.0,
                   time_warp=1.0, formant_scale=1.0):
    """Returns (x, f0, timeaxis): float64 waveform, per-frame f0 and frame times."""
    rng = np.random.default_rng(seed)
    n = int(round(fs * seconds * time_warp))
    t = np.arange(n) / fs
    tw = t / time_warp                      # position on the unwarped schedule
    f0_t = f0_base * 2.0 ** (0.4 * np.sin(2 * np.pi * 0.7 * tw + rng.uniform(0, 2 * np.pi)))
    voiced_t = (np.mod(tw, 0.5) < 0.40)
    phase = 2 * np.pi * np.cumsum(f0_t) / fs

    # slowly varying 4-formant envelope
    f_c = np.array([700.0, 1220.0, 2600.0, 3500.0]) * formant_scale
    f_bw = np.array([130.0, 170.0, 250.0, 350.0])
    gain = np.array([1.0, 0.6, 0.35, 0.2])
    wob = 1.0 + 0.08 * np.sin(2 * np.pi * 0.3 * tw[None, :] + np.arange(4)[:, None] * 1.3
                               + rng.uniform(0, 2 * np.pi, (4, 1)))
    centres = f_c[:, None] * wob            # (4, n)

    def envelope(freq):                      # freq: (n,) -> amplitude (n,)
        a = np.zeros_like(freq)
        for i in range(4):
            a += gain[i] / (1.0 + ((freq - centres[i]) / f_bw[i]) ** 2)
        return a + 0.002

    x = np.zeros(n)
    z1 = np.exp(1j * phase)
    z = np.ones(n, dtype=np.complex128)
    max_h = int((fs / 2) / (f0_base * 2.0 ** -0.4))
    for h in range(1, max_h + 1):
        z = z * z1
        fh = h * f0_t
        ok = fh < fs / 2 - 50.0
        if not ok.any():
            break
        x += np.where(ok, envelope(fh) * z.imag, 0.0)
    x *= voiced_t
    x /= max(np.abs(x).max(), 1e-12)
    noise = rng.standard_normal(n)
    x = x + noise * np.where(voiced_t, 10 ** (-35 / 20), 10 ** (-20 / 20))
    x *= 0.6 / np.abs(x).max()
    x = np.ascontiguousarray(x, dtype=np.float64)

    T = n * 1000 // fs // int(frame_period) + 1
    timeaxis = np.arange(T) * (frame_period / 1000.0)
    idx = np.minimum((timeaxis * fs).astype(np.int64), n - 1)
    f0 = np.where(voiced_t[idx], f0_t[idx], 0.0)
    return x, np.ascontiguousarray(f0), np.ascontiguousarray(timeaxis)
