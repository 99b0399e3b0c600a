"""The post-step of synthesised waveforms and their 16-bit PCM on the device (kwy_finish_pcm16_batch_dev): what
`Synthesizer.synthesize(normalize=True)` followed by `Wavdata.save(normalize=True)` computes on the host per file
(/root/reference/kwiiyatta/vocoder/abc/synthesizer.py:11-20, wavfile.py:8-29), for a batch of waveforms in HBM."""
import numpy as np

from .. import _lib
from .._lib import lib


def ceiling(peak_lv):
    """normalize_data's peak ceiling (the reference's power-style dB)"""
    return float(np.power(10, peak_lv / 10))


def pcm16_batch_dev(ctx, waves, frame_lens, fs, pcm_out, normalize_synth=True, synth_peak_lv=-1, normalize_save=True,
                    save_peak_lv=-1):
    """waves: float64 device tensors (kept as they are); frame_lens: the frame count of the feature each was rendered
    from; pcm_out: int16 device tensors of the same lengths, written.  save_peak_lv=None: mean removal only.
    Enqueued on the context's stream, not synchronised."""
    rows = [(y, y.numel(), int(T), p) for y, T, p in zip(waves, frame_lens, pcm_out)]
    jobs = _lib.job_array(_lib.FinishJob, rows)
    _lib.check(ctx, lib.kwy_finish_pcm16_batch_dev(
        ctx.handle, jobs, len(rows), int(fs), int(bool(normalize_synth)), ceiling(synth_peak_lv),
        int(bool(normalize_save)), float('nan') if save_peak_lv is None else ceiling(save_peak_lv)))
