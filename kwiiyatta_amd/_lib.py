"""ctypes binding of ``libkwy.so`` (include/kwy.h), the gfx950 HIP library.

There is no CPU fallback: importing this module without the built shared
object raises ImportError, and creating a context without a HIP device raises
RuntimeError (``libkwy: no HIP device available``).
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libkwy.so')

if not os.path.exists(_SO):
    raise ImportError(
        f'{_SO} is missing: build the HIP extension first '
        f'(python -c "import __graft_entry__ as g; g.build()" or kwiiyatta_amd/csrc/build.sh). '
        f'kwiiyatta_amd has no CPU fallback.')



def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7; the image also has
    /opt/rocm's.  Both have the same SONAME, so whichever is loaded first serves
    the whole process -- and torch only works with its own.  Load torch's copy
    first (without importing torch) so that libkwy.so and torch share one HIP
    runtime regardless of import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


_preload_hip_runtime()
lib = ctypes.CDLL(_SO)

c_dp = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double

KWY_OK, KWY_EINVAL, KWY_EHIP, KWY_ENOMEM, KWY_ENODEV, KWY_ENUMERIC = 0, -1, -2, -3, -4, -5

# name -> (restype, argtypes); mirrors include/kwy.h line by line
SIGNATURES = {
    'kwy_version': (c_int, []),
    'kwy_ctx_create': (c_int, [c_int, c_vp, ctypes.POINTER(c_vp)]),
    'kwy_ctx_destroy': (None, [c_vp]),
    'kwy_ctx_sync': (c_int, [c_vp]),
    'kwy_ctx_stream': (c_vp, [c_vp]),
    'kwy_ctx_arena_generation': (c_i64, [c_vp]),
    'kwy_ctx_reserve': (c_int, [c_vp, c_i64]),
    'kwy_copy_dev': (c_int, [c_vp, c_vp, c_vp, c_i64]),
    'kwy_ctx_set_randn_limit': (c_i64, [c_vp, c_i64]),
    'kwy_randn_stream': (c_int, [c_vp, c_i64, c_i64, c_vp]),
    'kwy_ctx_profile': (c_int, [c_vp, c_int]),
    'kwy_ctx_profile_read': (c_int, [c_vp, ctypes.c_char_p, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64)]),
    'kwy_last_error': (ctypes.c_char_p, [c_vp]),
    'kwy_create_error': (ctypes.c_char_p, []),
    'kwy_cheaptrick_fft_size': (c_int, [c_int, c_dbl]),
    'kwy_dio_frames': (c_i64, [c_int, c_i64, c_dbl]),
    'kwy_synth_length': (c_i64, [c_i64, c_dbl, c_int]),
    'kwy_cheaptrick': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_dbl, c_dbl, c_int,
                               c_dbl, c_vp]),
    'kwy_cheaptrick_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_dbl, c_dbl,
                                   c_int, c_dbl, c_vp]),
    'kwy_cheaptrick_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl]),
    'kwy_cheaptrick_mcep_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl, c_int, c_dbl]),
    'kwy_d4c_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_int]),
    'kwy_d4c': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_dbl, c_int, c_vp]),
    'kwy_d4c_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_dbl, c_int, c_vp]),
    'kwy_dio': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_dbl,
                        c_vp, c_vp]),
    'kwy_stonemask': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp]),
    'kwy_dio_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_dbl, c_vp, c_vp, c_vp]),
    'kwy_dio_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_dbl]),
    'kwy_stonemask_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp]),
    'kwy_stonemask_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int]),
    'kwy_finish_pcm16_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_dbl, c_int, c_dbl]),
    'kwy_finish_scratch_bytes': (c_i64, [c_i64]),
    'kwy_synthesize': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_dbl, c_int, c_dbl, c_i64,
                               c_vp]),
    'kwy_synthesize_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_dbl, c_int, c_dbl,
                                   c_i64, c_vp]),
    'kwy_synth_plan_bytes': (c_i64, [c_i64]),
    'kwy_synth_plan_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_int, c_i64, c_vp]),
    'kwy_synth_plan_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_int]),
    'kwy_synth_render_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_dbl, c_int, c_dbl, c_i64, c_vp]),
    'kwy_synth_render_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_int, c_dbl]),
    'kwy_sp2mc': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_dbl, c_vp]),
    'kwy_sp2mc_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_dbl, c_vp]),
    'kwy_mc2sp': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_int, c_vp]),
    'kwy_mc2sp_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_int, c_vp]),
    'kwy_fastdtw': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'kwy_fastdtw_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'kwy_fastdtw_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_int]),
    'kwy_align_features_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_dbl, c_dbl, c_dbl, c_vp]),
    'kwy_align_features_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_dbl]),
    'kwy_align_project_dev': (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp]),
    'kwy_align_project_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int]),
    'kwy_gather_rows_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp]),
    'kwy_gather_rows_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int]),
    'kwy_gmm_em_estep_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'kwy_gmm_em_sums_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    'kwy_gmm_em_means_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    'kwy_gmm_em_cov_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'kwy_gmm_em_cov_stats_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'kwy_gmm_em_finalize_dev': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_dbl, c_vp, c_vp]),
    'kwy_gmm_fit_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_dbl, c_dbl, ctypes.c_uint32, c_vp, c_vp, c_vp,
                                ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'kwy_gmm_em_scratch_bytes': (c_int, [c_i64, c_int, c_int, ctypes.POINTER(c_i64)]),
    'kwy_gmm_fit_comm_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_dbl, c_dbl, ctypes.c_uint32, c_vp, c_vp, c_vp, c_vp,
                                     ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'kwy_trim_length_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_vp]),
    'kwy_trim_length_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl]),
    'kwy_train_pad_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int]),
    'kwy_train_rows_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_i64, c_vp]),
    'kwy_is_voiced_dev': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    'kwy_align_even_dev': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64, c_int,
                                   c_vp, c_vp, c_i64, c_vp]),
    'kwy_delta_features_dev': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    'kwy_joint_rows_dev': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_dbl, c_vp, c_vp]),
    'kwy_km_colstats_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    'kwy_km_center_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    'kwy_km_pp_dist_dev': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    'kwy_km_chunks': (c_i64, [c_i64]),
    'kwy_km_pp_total_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'kwy_km_pp_pick_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'kwy_km_assign_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    'kwy_km_update_dev': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'kwy_km_onehot_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'kwy_km_lloyd_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_int, c_i64,
                         c_vp, c_vp]),
    'kwy_gmm_mlpg': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    'kwy_gmm_mlpg_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    'kwy_gmm_convert_frames': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    'kwy_gmm_convert_frames_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    'kwy_gmm_model_doubles': (c_i64, [c_int, c_int]),
    'kwy_aperiodicity_bands': (c_int, [c_int]),
    'kwy_code_aperiodicity': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    'kwy_code_aperiodicity_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    'kwy_decode_aperiodicity': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]),
    'kwy_decode_aperiodicity_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]),
    'kwy_stretch_log': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    'kwy_stretch_log_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    'kwy_np_state_bytes': (c_i64, []),
    'kwy_np_normal': (c_int, [c_vp, c_vp, c_dbl, c_dbl, c_int, c_i64, c_vp]),
    'kwy_np_normal_dev': (c_int, [c_vp, c_vp, c_dbl, c_dbl, c_int, c_i64, c_vp]),
    'kwy_np_normal_blocks_dev': (c_int, [c_vp, c_vp, c_dbl, c_dbl, c_int, c_int, c_i64, ctypes.POINTER(c_vp)]),
    'kwy_mc2b': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_vp]),
    'kwy_mc2b_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_dbl, c_vp]),
    'kwy_mlsa_synthesis': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_dbl, c_int, c_int, c_vp]),
    'kwy_mlsa_synthesis_dev': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_dbl, c_int, c_int, c_vp]),
    'kwy_mlsa_filter_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_dbl, c_int, c_int, c_int]),
    'kwy_gmm_prepare_dev': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'kwy_gmm_mlpg_model_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    'kwy_convert_mcep_dev': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    'kwy_convert_mcep_batch_dev': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _f = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _f.restype = _res
    _f.argtypes = _args


class Utterance(ctypes.Structure):
    """kwy_utterance (include/kwy.h): device pointers of one utterance for the batched analysis calls"""
    _fields_ = [('x', c_vp), ('x_length', c_i64), ('temporal_positions', c_vp), ('f0', c_vp), ('f0_length', c_i64),
                ('out', c_vp)]


def utterance_array(items):
    """items: (x, t, f0, out) device tensors per utterance -> a ctypes array of kwy_utterance"""
    arr = (Utterance * len(items))()
    for q, (x, t, f0, out) in zip(arr, items):
        q.x, q.x_length = x.data_ptr(), x.numel()
        q.temporal_positions, q.f0, q.f0_length = t.data_ptr(), f0.data_ptr(), f0.numel()
        q.out = out.data_ptr()
    return arr


class SynthJob(ctypes.Structure):
    """kwy_synth_job (include/kwy.h): one utterance of a batched rendering call"""
    _fields_ = [('plan', c_vp), ('spectrogram', c_vp), ('aperiodicity', c_vp), ('f0_length', c_i64),
                ('y_length', c_i64), ('y', c_vp)]


def synth_job_array(items):
    """items: (plan, sp, ap, y) device tensors per utterance (sp, ap: T x K; y: the waveform) -> a ctypes array of
    kwy_synth_job"""
    arr = (SynthJob * len(items))()
    for q, (plan, sp, ap, y) in zip(arr, items):
        q.plan, q.spectrogram, q.aperiodicity = plan.data_ptr(), sp.data_ptr(), ap.data_ptr()
        q.f0_length, q.y_length, q.y = sp.shape[0], y.numel(), y.data_ptr()
    return arr


def _job_struct(name, doc, fields):
    return type(name, (ctypes.Structure,), {'__doc__': doc, '_fields_': fields})


DtwJob = _job_struct('DtwJob', 'kwy_dtw_job (include/kwy.h): one pair of a batched FastDTW call',
                     [('x', c_vp), ('x_length', c_i64), ('y', c_vp), ('y_length', c_i64), ('dist', c_vp), ('path', c_vp),
                      ('path_len', c_vp)])
AlignJob = _job_struct('AlignJob', 'kwy_align_job: DTW feature rows of one utterance',
                       [('mc', c_vp), ('f0', c_vp), ('T', c_i64), ('out', c_vp)])
ProjectJob = _job_struct('ProjectJob', 'kwy_project_job: one path projected onto its target axis',
                         [('path', c_vp), ('path_len', c_vp), ('idx', c_vp), ('idx_capacity', c_i64), ('n_out', c_vp)])
GatherJob = _job_struct('GatherJob', 'kwy_gather_job: one row gather',
                        [('src', c_vp), ('src_rows', c_i64), ('idx', c_vp), ('n', c_i64), ('dst', c_vp)])
ConvertJob = _job_struct('ConvertJob', 'kwy_convert_job: one utterance of a batched conversion',
                         [('mc', c_vp), ('T', c_i64), ('mc_out', c_vp)])
TrimJob = _job_struct('TrimJob', 'kwy_trim_job', [('sp', c_vp), ('T', c_i64), ('n_out', c_vp)])
PadJob = _job_struct('PadJob', 'kwy_pad_job', [('f0', c_vp), ('n', c_i64), ('f0_pad', c_vp), ('ap_pad', c_vp), ('voiced', c_vp)])
TrainJob = _job_struct('TrainJob', 'kwy_train_job',
                       [('path', c_vp), ('path_len', c_vp), ('feat_x', c_vp), ('feat_y', c_vp), ('mc_x', c_vp), ('mc_y', c_vp),
                        ('x_length', c_i64), ('y_length', c_i64), ('n_rows', c_vp)])
F0Job = _job_struct('F0Job', 'kwy_f0_job: the DIO f0 track of one utterance',
                   [('x', c_vp), ('x_length', c_i64), ('temporal_positions', c_vp), ('f0', c_vp), ('status', c_vp)])
MlsaJob = _job_struct('MlsaJob', 'kwy_mlsa_job: one signal through the MLSA filter of its mel-cepstra',
                      [('x', c_vp), ('x_length', c_i64), ('mc', c_vp), ('T', c_i64), ('y', c_vp)])
FinishJob = _job_struct('FinishJob', 'kwy_finish_job: post-step + 16-bit PCM of one synthesised waveform',
                        [('y', c_vp), ('y_length', c_i64), ('frame_len', c_i64), ('pcm', c_vp)])
SynthPlanJob = _job_struct('SynthPlanJob', 'kwy_synth_plan_job: the pulse placement of one utterance',
                           [('f0', c_vp), ('f0_length', c_i64), ('y_length', c_i64), ('plan', c_vp)])


def job_array(struct, rows):
    """rows: tuples in the struct's field order; device tensors become their data pointers"""
    arr = (struct * len(rows))()
    for q, row in zip(arr, rows):
        for (name, _), v in zip(struct._fields_, row):
            setattr(q, name, v.data_ptr() if hasattr(v, 'data_ptr') else v)
    return arr


class Context:
    """One HIP stream + device scratch (``kwy_ctx``).  Calls on one context are
    serialised; use one context per thread / utterance stream."""

    def __init__(self, device=0, stream=None):
        h = c_vp()
        rc = lib.kwy_ctx_create(int(device), c_vp(stream) if stream else None, ctypes.byref(h))
        if rc != KWY_OK:
            msg = lib.kwy_create_error().decode()
            raise RuntimeError(msg or f'kwy_ctx_create failed ({rc})')
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, 'handle', None):
            lib.kwy_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self, lib.kwy_ctx_sync(self.handle))

    def profile(self, enable=True):
        check(self, lib.kwy_ctx_profile(self.handle, int(bool(enable))))

    def set_randn_limit(self, draws=-1):
        """use only the first `draws` entries of the device's randn table (beyond: jump-ahead); -1 = all"""
        return int(lib.kwy_ctx_set_randn_limit(self.handle, int(draws)))

    def arena_generation(self):
        """changes whenever the scratch arena is relocated (captured HIP graphs of this context become invalid)"""
        return int(lib.kwy_ctx_arena_generation(self.handle))

    def profile_read(self, kernel):
        """(total_ms, launches) of `kernel` since the last read; synchronises the stream."""
        ms, n = c_dbl(), c_i64()
        check(self, lib.kwy_ctx_profile_read(self.handle, kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def error(self):
        return lib.kwy_last_error(self.handle).decode()


def check(ctx, rc):
    if rc == KWY_OK:
        return
    msg = ctx.error() if ctx is not None else ''
    if rc in (KWY_EINVAL, KWY_ENUMERIC):
        raise ValueError(msg or 'invalid argument')
    if rc == KWY_ENOMEM:
        raise MemoryError(msg or 'device out of memory')
    raise RuntimeError(msg or f'libkwy error {rc}')


_tls = threading.local()
_default_device = 0


def set_default_device(device):
    global _default_device
    _default_device = int(device)
    _tls.ctx = None


def default_context():
    """Per-thread default context on the default device (LOCAL_RANK-agnostic;
    multi-GPU drivers call set_default_device(local_rank))."""
    ctx = getattr(_tls, 'ctx', None)
    if ctx is None or ctx.handle is None or ctx.device != _default_device:
        ctx = Context(_default_device)
        _tls.ctx = ctx
    return ctx


def as_f64(a, name='ndarray'):
    """pyworld's input contract (reference tests/kwiiyatta/vocoder/test_world.py:21-40)."""
    a = np.asarray(a)
    if not a.flags['C_CONTIGUOUS']:
        raise ValueError('ndarray is not C-contiguous')
    if a.dtype != np.float64:
        raise ValueError(f"Buffer dtype mismatch, expected 'double' but got {a.dtype!s}")
    return a


def ptr(a):
    return c_vp(a.ctypes.data)
