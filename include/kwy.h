/*
 * include/kwy.h -- C ABI of libkwy.so, the MI355X (gfx950) implementation of
 * kwiiyatta's per-utterance conversion hot path.
 *
 * The reference (Iselix/kwiiyatta) has no native code: its replaceable seam is
 * the set of third-party calls it makes.  Every entry
 * point below replaces one of those calls; the comment on each names the
 * reference call site.  Conventions:
 *
 *   - all arrays are float64, C-contiguous, caller-allocated; the library never
 *     retains or frees a caller pointer
 *   - `kwy_<op>`      takes HOST pointers (what a ctypes/numpy binding passes);
 *                     it stages through HBM, runs the HIP kernels and copies
 *                     the result back (synchronous on return)
 *   - `kwy_<op>_dev`  takes DEVICE pointers; work is enqueued on the context's
 *                     HIP stream and NOT synchronised (call kwy_ctx_sync)
 *   - return value: 0 on success, a negative KWY_E* code otherwise;
 *     kwy_last_error(ctx) gives a message.  No C++ exception crosses the ABI.
 *   - a kwy_ctx owns one HIP stream plus its device scratch; calls on one
 *     context are serialised, different contexts run concurrently
 *     (one context per utterance stream / per Python thread).
 *   - There is NO CPU fallback: without a HIP device kwy_ctx_create fails.
 */
#ifndef KWY_H_
#define KWY_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KWY_OK 0
#define KWY_EINVAL (-1)   /* bad argument                               -> ValueError   */
#define KWY_EHIP (-2)     /* HIP runtime / launch failure               -> RuntimeError */
#define KWY_ENOMEM (-3)   /* device allocation failed                   -> MemoryError  */
#define KWY_ENODEV (-4)   /* no usable HIP device                       -> RuntimeError */
#define KWY_ENUMERIC (-5) /* numerical failure (e.g. covariance not PD) -> ValueError   */

typedef struct kwy_ctx kwy_ctx;

/* ---- context --------------------------------------------------------------- */
int kwy_version(void);
/* device: HIP device ordinal.  stream: an existing hipStream_t to enqueue on
 * (e.g. torch's current stream), or NULL to let the context create its own. */
int kwy_ctx_create(int device, void *stream, kwy_ctx **out);
void kwy_ctx_destroy(kwy_ctx *ctx);
int kwy_ctx_sync(kwy_ctx *ctx);
void *kwy_ctx_stream(kwy_ctx *ctx);
/* The context's scratch arena grows on demand, and growing RELOCATES it (the stream is synchronised first).  A HIP
 * graph captured from calls on this context holds arena addresses: it is valid only while
 * kwy_ctx_arena_generation() returns the value it had at capture time.  kwy_ctx_reserve() grows the arena to at
 * least `bytes` ahead of time (e.g. for the longest utterance of a batch) so that later calls do not move it. */
int64_t kwy_ctx_arena_generation(kwy_ctx *ctx);
int kwy_ctx_reserve(kwy_ctx *ctx, int64_t bytes);
/* dst[0 .. bytes) <- src[0 .. bytes), device to device, as a KERNEL on the context's stream (bytes a multiple of 8;
 * the buffers must not overlap).  For drivers that move a step's inputs and results between staging blocks and the
 * buffers a captured graph reads (Wavdata -> Analyzer's copy, kwiiyatta/vocoder/world.py:14-17): a copy-engine
 * transfer per stream and direction does not scale over many streams here, a kernel does. */
int kwy_copy_dev(kwy_ctx *ctx, void *dst, const void *src, int64_t bytes);
/* WORLD's analysis / synthesis noise is ONE fixed pseudo-random sequence (the generator is reseeded at the entry of
 * every pyworld call); the library keeps its first 2^KWY_RANDN_LOG2 draws (environment, default 25 = 128 MB, 0 = none)
 * in a table per device and computes draws beyond it by jump-ahead -- the same numbers either way.  This call lowers
 * the number of table draws THIS context uses (draws < 0: all of them); returns the length now in use.  For tests of
 * the jump-ahead path. */
int64_t kwy_ctx_set_randn_limit(kwy_ctx *ctx, int64_t draws);
/* draws [first, first + count) of that sequence as WORLD's randn() returns them (host array), read from the table */
int kwy_randn_stream(kwy_ctx *ctx, int64_t first, int64_t count, double *out);
/* Per-kernel timing: when enabled, the main kernels are bracketed by HIP events
 * on the context's stream.  kwy_ctx_profile_read synchronises the stream and
 * returns the summed duration [ms] and launch count of `kernel` since the last
 * read (kernel names as in rocprofv3, without template arguments, e.g.
 * "k_d4c_body"). */
int kwy_ctx_profile(kwy_ctx *ctx, int enable);
int kwy_ctx_profile_read(kwy_ctx *ctx, const char *kernel, double *total_ms, int64_t *count);
const char *kwy_last_error(kwy_ctx *ctx);
/* error text when kwy_ctx_create itself failed (no context to ask) */
const char *kwy_create_error(void);

/* ---- sizes (pure host arithmetic, no device needed) -------------------------- */
/* pyworld.get_cheaptrick_fft_size(fs, f0_floor)      kwiiyatta/vocoder/world.py:96 */
int kwy_cheaptrick_fft_size(int fs, double f0_floor);
/* number of frames pyworld.dio returns               kwiiyatta/vocoder/world.py:35 */
int64_t kwy_dio_frames(int fs, int64_t x_length, double frame_period_ms);
/* output length of pyworld.synthesize                kwiiyatta/vocoder/world.py:86 */
int64_t kwy_synth_length(int64_t f0_length, double frame_period_ms, int fs);

/* ---- WORLD analysis ----------------------------------------------------------- */
/* pyworld.cheaptrick(x, f0, t, fs, q1, f0_floor, fft_size)
 *                                                    kwiiyatta/vocoder/world.py:45
 * fft_size <= 0: derive from (fs, f0_floor).  out: T x (fft_size/2+1).
 * out_div: the result is divided by this value (pass fs to fold the
 * `spectrum_envelope /= fs` of world.py:50 into the kernel; 1.0 = pyworld). */
int kwy_cheaptrick(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                   const double *temporal_positions, const double *f0, int64_t f0_length,
                   double q1, double f0_floor, int fft_size, double out_div, double *out);
int kwy_cheaptrick_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                       const double *temporal_positions, const double *f0, int64_t f0_length,
                       double q1, double f0_floor, int fft_size, double out_div, double *out);

/* The same call for a batch of utterances of one sampling rate (device pointers), one grid over all their frames:
 * what Analyzer.extract_spectrum_envelope does file after file (kwiiyatta/vocoder/world.py:43-52) for both sides of a
 * pair, or for every file of a corpus.  A lone utterance's ~2 000 workgroups are 2.2 rounds of what the chip holds. */
typedef struct kwy_utterance {
  const double *x;                   /* waveform, x_length samples */
  int64_t x_length;
  const double *temporal_positions;  /* f0_length frame times [s] */
  const double *f0;                  /* f0_length */
  int64_t f0_length;
  double *out;                       /* f0_length x (fft_size/2+1) */
} kwy_utterance;
int kwy_cheaptrick_batch_dev(kwy_ctx *ctx, const kwy_utterance *utterances, int count, int fs, double q1,
                             double f0_floor, int fft_size, double out_div);
/* CheapTrick and sp2mc in one, for consumers that only need the mel-cepstrum (the alignment, the conversion):
 *   Analyzer.extract_spectrum_envelope + MelCepstrum of it        kwiiyatta/vocoder/world.py:43-52, mcep.py:68-71
 * CheapTrick's last steps are: lifter the cepstrum, transform back, exp; sp2mc's first: log, transform to the
 * cepstrum.  The fused form keeps the liftered cepstrum and applies pysptk's frequency transform to it directly (an
 * f64 matrix product): no K-bin envelope row is written or read.  utterances[i].out: f0_length x (order + 1)
 * mel-cepstral coefficients, equal to kwy_sp2mc_dev(kwy_cheaptrick_dev(...)) up to the rounding of the skipped
 * exp / log round trip (1e-12 relative in the tests).  order <= 63. */
int kwy_cheaptrick_mcep_batch_dev(kwy_ctx *ctx, const kwy_utterance *utterances, int count, int fs, double q1,
                                  double f0_floor, int fft_size, double out_div, int order, double alpha);

/* pyworld.d4c(x, f0, t, fs, threshold, fft_size)     kwiiyatta/vocoder/world.py:55
 * out: T x (fft_size/2+1). */
int kwy_d4c(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
            const double *temporal_positions, const double *f0, int64_t f0_length,
            double threshold, int fft_size, double *out);
int kwy_d4c_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                const double *temporal_positions, const double *f0, int64_t f0_length,
                double threshold, int fft_size, double *out);
/* ... for a batch of utterances (see kwy_cheaptrick_batch_dev) */
int kwy_d4c_batch_dev(kwy_ctx *ctx, const kwy_utterance *utterances, int count, int fs, double threshold,
                      int fft_size);

/* pyworld.dio(x, fs, f0_floor, f0_ceil, channels_in_octave, frame_period, speed,
 *             allowed_range) -> (f0, t)              kwiiyatta/vocoder/world.py:35
 * only speed == 1 (pyworld's default, the only value kwiiyatta uses). */
int kwy_dio(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, double f0_floor,
            double f0_ceil, double channels_in_octave, double frame_period_ms, int speed,
            double allowed_range, double *temporal_positions, double *f0);
/* pyworld.stonemask(x, f0, t, fs)                    kwiiyatta/vocoder/world.py:39 */
int kwy_stonemask(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                  const double *temporal_positions, const double *f0, int64_t f0_length,
                  double *refined_f0);
/* The f0 track without the host (round 5): what Analyzer.extract_f0 does in front of every analysis,
 * kwiiyatta/vocoder/world.py:33-41, on DEVICE pointers, enqueued on the context's stream and not synchronised; the
 * batch forms take the utterances of a wave in one pass of launches (DIO <= 32 per pass: the two sides of a wave of 16
 * pairs share the single-wavefront contour repair; StoneMask <= 16), as kwy_cheaptrick_batch_dev does.  Every job's result equals the host entry's bit for bit (the host entries run the same pass on staged copies).
 * status (one int32 on the device per utterance, or NULL): cleared by the call, set to 1 when an engine's
 * zero-crossing buffer overflowed (kwy_dio reports that as KWY_EHIP); a driver reads the words back once per wave. */
typedef struct kwy_f0_job {
  const double *x;               /* waveform, x_length samples */
  int64_t x_length;
  double *temporal_positions;    /* kwy_dio_frames(fs, x_length, frame_period_ms) frame times [s], written */
  double *f0;                    /* as many f0 values, written */
  int32_t *status;               /* see above; may be NULL */
} kwy_f0_job;
int kwy_dio_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, double f0_floor, double f0_ceil,
                double channels_in_octave, double frame_period_ms, int speed, double allowed_range,
                double *temporal_positions, double *f0, int32_t *status);
int kwy_dio_batch_dev(kwy_ctx *ctx, const kwy_f0_job *jobs, int count, int fs, double f0_floor, double f0_ceil,
                      double channels_in_octave, double frame_period_ms, int speed, double allowed_range);
int kwy_stonemask_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                      const double *temporal_positions, const double *f0, int64_t f0_length,
                      double *refined_f0);
/* utterances[i].out: the refined f0 (f0_length values); .f0: the initial track */
int kwy_stonemask_batch_dev(kwy_ctx *ctx, const kwy_utterance *utterances, int count, int fs);

/* ---- WORLD synthesis ------------------------------------------------------------ */
/* pyworld.synthesize(f0, sp, ap, fs, frame_period)   kwiiyatta/vocoder/world.py:86-92
 * sp_mul: every spectrogram value is multiplied by this before use (pass fs to
 * fold `spectrum_envelope * fs` of world.py:88; 1.0 = pyworld).
 * y: kwy_synth_length(...) samples. */
int kwy_synthesize(kwy_ctx *ctx, const double *f0, int64_t f0_length, const double *sp,
                   const double *ap, int fft_size, double frame_period_ms, int fs,
                   double sp_mul, int64_t y_length, double *y);
int kwy_synthesize_dev(kwy_ctx *ctx, const double *f0, int64_t f0_length, const double *sp,
                       const double *ap, int fft_size, double frame_period_ms, int fs,
                       double sp_mul, int64_t y_length, double *y);
/* The same call in two steps, for pipelines: the pulse placement depends on f0 alone (WORLD's phase accumulation and
 * zero-crossing search, synthesis.cpp GetTimeBase [RECALL]), so it can run on another context / stream while the
 * spectral features are still being computed.  plan: kwy_synth_plan_bytes(y_length) bytes of device memory, written
 * by kwy_synth_plan_dev and read by kwy_synth_render_dev (same f0-derived arguments in both calls). */
int64_t kwy_synth_plan_bytes(int64_t y_length);
int kwy_synth_plan_dev(kwy_ctx *ctx, const double *f0, int64_t f0_length, int fft_size, double frame_period_ms,
                       int fs, int64_t y_length, void *plan);
int kwy_synth_render_dev(kwy_ctx *ctx, const void *plan, int64_t f0_length, const double *sp, const double *ap,
                         int fft_size, double frame_period_ms, int fs, double sp_mul, int64_t y_length, double *y);
/* The pulse placement of several utterances in one pass of launches: the serial phase chain of an utterance is one
 * workgroup, so N utterances occupy N compute units side by side (Synthesizer.synthesize runs file after file,
 * kwiiyatta/vocoder/world.py:80-92).  Every plan equals kwy_synth_plan_dev's bit for bit. */
typedef struct kwy_synth_plan_job {
  const double *f0;      /* f0_length */
  int64_t f0_length;
  int64_t y_length;
  void *plan;            /* kwy_synth_plan_bytes(y_length) bytes */
} kwy_synth_plan_job;
int kwy_synth_plan_batch_dev(kwy_ctx *ctx, const kwy_synth_plan_job *jobs, int count, int fft_size,
                             double frame_period_ms, int fs);
/* ... for a batch of utterances: one pass of launches renders the pulses of all of them (every utterance a slice of
 * the grids), as kwy_cheaptrick_batch_dev / kwy_d4c_batch_dev analyse them.  What Synthesizer.synthesize does file
 * after file (kwiiyatta/vocoder/world.py:63-80, resynthesize_voice.py:46-79).  Every job's waveform equals
 * kwy_synth_render_dev's bit for bit. */
typedef struct kwy_synth_job {
  const void *plan;             /* kwy_synth_plan_dev's output for this utterance */
  const double *spectrogram;    /* f0_length x (fft_size/2+1) */
  const double *aperiodicity;   /* f0_length x (fft_size/2+1) */
  int64_t f0_length;
  int64_t y_length;
  double *y;                    /* y_length samples */
} kwy_synth_job;
int kwy_synth_render_batch_dev(kwy_ctx *ctx, const kwy_synth_job *jobs, int count, int fft_size,
                               double frame_period_ms, int fs, double sp_mul);

/* The post-step of a synthesised waveform and its 16-bit PCM, for a batch of waveforms in device memory:
 *   Synthesizer.synthesize(normalize=True): wavdata.normalize(None); normalize_data() on the first `frame_len`
 *     1 ms pieces                                     kwiiyatta/vocoder/abc/synthesizer.py:11-20
 *   Wavdata.save(normalize=True): data -= mean; normalize_data(data, peak_lv); (data * 2**15).astype(np.int16)
 *                                                     kwiiyatta/wavfile.py:8-29
 * normalize_synth / normalize_save: the two `normalize` flags; piece_ceiling / save_ceiling: 10 ** (peak_lv / 10) of
 * the two calls (save_ceiling NaN: peak_lv=None, mean removal only).  The samples equal the host path's exactly
 * (numpy's chunked pairwise mean is reproduced).  y is not modified; a wave downloads 2 bytes per sample.  Enqueued on
 * the context's stream, not synchronised.  KWY_EINVAL where the reference's loop would fail (a piece beyond the end). */
typedef struct kwy_finish_job {
  const double *y;      /* y_length samples (kwy_synthesize's output) */
  int64_t y_length;
  int64_t frame_len;    /* frames of the feature the waveform was rendered from */
  int16_t *pcm;         /* y_length samples, written */
} kwy_finish_job;
int kwy_finish_pcm16_batch_dev(kwy_ctx *ctx, const kwy_finish_job *jobs, int count, int fs, int normalize_synth,
                               double piece_ceiling, int normalize_save, double save_ceiling);
/* bytes of the context's scratch arena one waveform of that length needs in the call above (kwy_ctx_reserve) */
int64_t kwy_finish_scratch_bytes(int64_t y_length);

/* ---- mel-cepstrum ---------------------------------------------------------------- */
/* pysptk.sp2mc(spec, order, alpha) row-wise          kwiiyatta/vocoder/mcep.py:71
 * sp: T x K (K = fftlen/2+1), mc: T x (order+1). */
int kwy_sp2mc(kwy_ctx *ctx, const double *sp, int64_t T, int K, int order, double alpha,
              double *mc);
int kwy_sp2mc_dev(kwy_ctx *ctx, const double *sp, int64_t T, int K, int order, double alpha,
                  double *mc);
/* pysptk.mc2sp(mc, alpha, fftlen) row-wise           kwiiyatta/vocoder/mcep.py:65 */
int kwy_mc2sp(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, int fftlen,
              double *sp);
int kwy_mc2sp_dev(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha,
                  int fftlen, double *sp);

/* ---- alignment -------------------------------------------------------------------- */
/* fastdtw.fastdtw(x, y, radius, dist=2) -> (dist, path)
 *                                                    kwiiyatta/vocoder/align.py:71
 * x: Tx x dim, y: Ty x dim; path: capacity (Tx+Ty) x 2 int32, *path_len pairs written.
 * radius >= 1 (KWY_EINVAL otherwise: fastdtw 0.3.2 raises a KeyError for radius 0). */
int kwy_fastdtw(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty,
                int dim, int radius, double *dist, int32_t *path, int64_t *path_len);
int kwy_fastdtw_dev(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty,
                    int dim, int radius, double *dist, int32_t *path, int64_t *path_len);
/* ... for a batch of pairs (device pointers): every kernel of the level recursion runs all pairs in one grid -- the
 * serial recurrence of a pair is one workgroup, so N pairs occupy N compute units instead of one after the other --
 * what align does pair after pair over a corpus (kwiiyatta/vocoder/align.py:61-96,123-131, convert_voice.py:35-46).
 * Every job's distance and path equal kwy_fastdtw_dev's bit for bit.  The scratch capacities are bounds for any
 * pair of lengths (no overflow, whatever the length ratio). */
typedef struct kwy_dtw_job {
  const double *x;       /* x_length x dim */
  int64_t x_length;
  const double *y;       /* y_length x dim */
  int64_t y_length;
  double *dist;          /* 1 */
  int32_t *path;         /* capacity (x_length + y_length) x 2 */
  int64_t *path_len;     /* 1 */
} kwy_dtw_job;
int kwy_fastdtw_batch_dev(kwy_ctx *ctx, const kwy_dtw_job *jobs, int count, int dim, int radius);

/* Device-side glue that keeps a source/target pair resident in HBM between the
 * stages (used by the batched pipeline; the host API does these in numpy):
 * DTW feature rows [power, voicing, mc1..mcN] -- make_feature(power='binalize',
 * power_pivot='max', vuv='f0')                      kwiiyatta/vocoder/align.py:20-58
 * mc: T x ncoef, out: T x (ncoef+1). */
int kwy_align_features_dev(kwy_ctx *ctx, const double *mc, int64_t T, int ncoef, const double *f0,
                           double power_weight, double power_threshold, double vuv_weight,
                           double *out);
typedef struct kwy_align_job {
  const double *mc;      /* T x ncoef */
  const double *f0;      /* T (or any per-frame voicing value: > 0 = voiced) */
  int64_t T;
  double *out;           /* T x (ncoef + 1) */
} kwy_align_job;
int kwy_align_features_batch_dev(kwy_ctx *ctx, const kwy_align_job *jobs, int count, int ncoef, double power_weight,
                                 double power_threshold, double vuv_weight);
/* project_path_iter(path, trim, trim_len): one x index per y frame
 *                                                    kwiiyatta/vocoder/align.py:99-120
 * path / path_len as produced by kwy_fastdtw_dev (device); trim_len = 0: no trimming. */
int kwy_align_project_dev(kwy_ctx *ctx, const int32_t *path, const int64_t *path_len, int trim_len,
                          int32_t *idx, int64_t idx_capacity, int64_t *n_out);
typedef struct kwy_project_job {
  const int32_t *path;
  const int64_t *path_len;
  int32_t *idx;
  int64_t idx_capacity;
  int64_t *n_out;
} kwy_project_job;
int kwy_align_project_batch_dev(kwy_ctx *ctx, const kwy_project_job *jobs, int count, int trim_len);
/* Feature.__getitem__(list): dst[i] = src[idx[i]]    kwiiyatta/vocoder/abc/feature.py:170-194 */
int kwy_gather_rows_dev(kwy_ctx *ctx, const double *src, int64_t src_rows, int width,
                        const int32_t *idx, int64_t n, double *dst);
typedef struct kwy_gather_job {
  const double *src;     /* src_rows x width */
  int64_t src_rows;
  const int32_t *idx;    /* n */
  int64_t n;
  double *dst;           /* n x width */
} kwy_gather_job;
int kwy_gather_rows_batch_dev(kwy_ctx *ctx, const kwy_gather_job *jobs, int count, int width);

/* ---- converter apply ---------------------------------------------------------------- */
/* delta_features(X, DELTA_WINDOWS) + MLPG(gmm, windows, diff).transform(X)[:, :d]
 *                         kwiiyatta/converter/delta.py:39-50, gmm.py:28-34
 * x: T x d static features.  GMM: M components over the joint (2*3d)-dim
 * static+delta+delta2 space (source | target), full covariances.
 * y: T x d converted static features. */
int kwy_gmm_mlpg(kwy_ctx *ctx, const double *x, int64_t T, int d, int M,
                 const double *weights, const double *means, const double *covs, int diff,
                 double *y);
int kwy_gmm_mlpg_dev(kwy_ctx *ctx, const double *x, int64_t T, int d, int M,
                     const double *weights, const double *means, const double *covs, int diff,
                     double *y);
/* MLPG(gmm, windows=DELTA_WINDOWS[0:1], diff).transform(X): GMMFeatureConverter.convert(feature, mlpg=False)
 *                                                    kwiiyatta/converter/gmm.py:28-34
 * frame-wise conversion without trajectory smoothing: y_t = sum_m p(m | x_t) E[y | x_t, m] (nnmnkwii MLPGBase).
 * x, y: T x D rows of whatever the mixture was trained on (D <= 128); the mixture is over 2 D joint dimensions. */
int kwy_gmm_convert_frames(kwy_ctx *ctx, const double *x, int64_t T, int D, int M, const double *weights,
                           const double *means, const double *covs, int diff, double *y);
int kwy_gmm_convert_frames_dev(kwy_ctx *ctx, const double *x, int64_t T, int D, int M, const double *weights,
                               const double *means, const double *covs, int diff, double *y);
/* The part of MLPG(gmm, windows, diff).__init__ that depends on the GMM only
 * (nnmnkwii computes it in the constructor, kwiiyatta/converter/gmm.py:32 builds one MLPG per
 * convert call): per mixture the Cholesky factor of Sxx, A = Syx Sxx^-1, b = mu_y - A mu_x, the
 * conditional variances and the log-density constants.  model: kwy_gmm_model_doubles(d, M)
 * doubles of device memory, filled once and reused by kwy_gmm_mlpg_model_dev for every utterance.
 * kwy_gmm_prepare_dev synchronises the stream (it reports a non-positive-definite Sxx). */
int64_t kwy_gmm_model_doubles(int d, int M);
int kwy_gmm_prepare_dev(kwy_ctx *ctx, const double *weights, const double *means, const double *covs,
                        int d, int M, int diff, double *model);
int kwy_gmm_mlpg_model_dev(kwy_ctx *ctx, const double *x, int64_t T, int d, int M, const double *model,
                           double *y);
/* MelCepstrumFeatureConverter.convert(mel_cepstrum) at the converter's sampling rate
 *                                                    kwiiyatta/converter/mcep.py:47-61
 * mc, mc_out: T x (d+1); column 0 (the power coefficient) is kept, columns 1..d are converted (delta features,
 * GMM, MLPG) with a prepared model (diff = 0 or 1 as given to kwy_gmm_prepare_dev). */
int kwy_convert_mcep_dev(kwy_ctx *ctx, const double *mc, int64_t T, int d, int M, const double *model,
                         double *mc_out);
/* ... for a batch of utterances: the frame-parallel kernels (deltas, mixture log-densities on the matrix cores,
 * conditional means) run over the frames of all of them, the trajectory solves of all of them share two launches
 * (convert_voice.py:35-46 converts file after file).  Every job's result equals kwy_convert_mcep_dev's bit for bit. */
typedef struct kwy_convert_job {
  const double *mc;      /* T x (d + 1) */
  int64_t T;
  double *mc_out;        /* T x (d + 1) */
} kwy_convert_job;
int kwy_convert_mcep_batch_dev(kwy_ctx *ctx, const kwy_convert_job *jobs, int count, int d, int M,
                               const double *model);

/* ---- cross-rate aperiodicity codec ------------------------------------------------------
 * pyworld.code_aperiodicity(ap, fs) / pyworld.decode_aperiodicity(coded, fs, fft_size)
 *                                        kwiiyatta/vocoder/world.py:98-145
 * ap: T x (fft_size/2+1); coded: T x kwy_aperiodicity_bands(fs) on the coding side.  The
 * decoder takes the number of coded columns explicitly (the reference truncates or extends the
 * coded array when it changes the sampling rate, world.py:121-143). */
int kwy_aperiodicity_bands(int fs);
int kwy_code_aperiodicity(kwy_ctx *ctx, const double *ap, int64_t T, int fs, int fft_size, double *coded);
int kwy_code_aperiodicity_dev(kwy_ctx *ctx, const double *ap, int64_t T, int fs, int fft_size, double *coded);
int kwy_decode_aperiodicity(kwy_ctx *ctx, const double *coded, int64_t T, int fs, int fft_size, int bands,
                            double *ap);
int kwy_decode_aperiodicity_dev(kwy_ctx *ctx, const double *coded, int64_t T, int fs, int fft_size, int bands,
                                double *ap);

/* ---- spectral-axis stretch (another number of bins at the same sampling rate) ------------------------
 * Synthesizer._reshape_feature on log values, as reshape_spectrum_envelope / reshape_aperiodicity call it:
 *   np.exp(scipy.signal.resample_poly(np.hstack((edge x pad, np.log(rows), edge x pad)), new_K, K, axis=1)[:, trim:-trim])
 *                                        kwiiyatta/vocoder/abc/synthesizer.py:31-54
 * rows: T x K (positive), out: T x new_K.  The FIR (scipy's firwin, Kaiser beta 5, 20 max(up, down) + 1 taps) is
 * designed by the library per (K, new_K) and cached in the context. */
int kwy_stretch_log(kwy_ctx *ctx, const double *rows, int64_t T, int K, int new_K, double *out);
int kwy_stretch_log_dev(kwy_ctx *ctx, const double *rows, int64_t T, int K, int new_K, double *out);

/* ---- numpy's legacy normal generator (the source of pad_silence's spectra) --------------------------------------
 * np.abs(np.random.normal(0, EPS / fs, (frame_len, spectrum_len)))
 *        WorldSynthesizer._silence_spectrum_envelope, kwiiyatta/vocoder/world.py:158-161 (via pad_silence,
 *        vocoder/abc/feature.py:19-41, for both sides of every aligned pair)
 * out[i] = loc + scale * g_i (its absolute value when take_abs), i < n, where g continues numpy's RandomState stream
 * from `state`: MT19937 + polar Box-Muller with the same accept / reject decisions and the same use of the cached
 * second value; `state` is advanced exactly as numpy advances it.  state: kwy_np_state_bytes() bytes laid out as
 * { uint32 key[624]; int32 pos; int32 has_gauss; double cached_gaussian } = numpy's get_state() tuple.  Values agree
 * with numpy's up to the rounding of log() (<= 1 ulp). */
int64_t kwy_np_state_bytes(void);
int kwy_np_normal(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n, double *out);
int kwy_np_normal_dev(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n, double *out);
/* `count` consecutive requests of n_each values each in one call -- the four pad blocks of one aligned pair (source
 * head, source tail, target head, target tail), or those of all pairs of a batch in pair order; outs: HOST array of
 * `count` device pointers */
int kwy_np_normal_blocks_dev(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int count,
                             int64_t n_each, double *const *outs);

/* ---- MLSA differential-spectrum filter ---------------------------------------------------
 * pysptk.mc2b(mc, alpha)                                     kwiiyatta/filter/mlsa.py:28
 * pysptk.synthesis.Synthesizer(MLSADF(order, alpha, pd), hopsize).synthesis(x, b)     :24-29
 * mc, b: T x (order+1).  Frame i filters samples [i hopsize, (i+1) hopsize) with coefficients
 * interpolated linearly from frame i-1's to frame i's (frame 0 from its own); the input is
 * scaled by exp(b[0]); frames that reach the end of x are left unprocessed (output 0 there;
 * upstream leaves them uninitialised).  pd: Pade order, 4 (pysptk default) or 5.  x, y: n samples. */
int kwy_mc2b(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, double *b);
int kwy_mc2b_dev(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, double *b);
int kwy_mlsa_synthesis(kwy_ctx *ctx, const double *x, int64_t n, const double *b, int64_t T, int order,
                       double alpha, int pd, int hopsize, double *y);
int kwy_mlsa_synthesis_dev(kwy_ctx *ctx, const double *x, int64_t n, const double *b, int64_t T, int order,
                           double alpha, int pd, int hopsize, double *y);
/* kwiiyatta.apply_mlsa_filter for a batch of signals (device pointers, not synchronised): what convert_voice.py writes
 * as <name>.diff.wav for every file (kwiiyatta/convert_voice.py:19,39-40; filter/mlsa.py:9-30) -- pysptk.mc2b of the
 * job's mel-cepstra (ignore_c0 != 0: with c0 taken as zero, filter/mlsa.py's `mcep.data[:, 0] = 0`), then the MLSA
 * filter over the waveform, one wavefront per signal and all signals of the batch side by side in one launch (the
 * recursion is serial in the sample: N signals occupy N compute units).  Every job's output equals kwy_mc2b_dev +
 * kwy_mlsa_synthesis_dev bit for bit. */
typedef struct kwy_mlsa_job {
  const double *x;      /* x_length samples in */
  int64_t x_length;
  const double *mc;     /* T x (order + 1) mel-cepstra */
  int64_t T;
  double *y;            /* x_length samples out */
} kwy_mlsa_job;
int kwy_mlsa_filter_batch_dev(kwy_ctx *ctx, const kwy_mlsa_job *jobs, int count, int order, double alpha, int pd,
                              int hopsize, int ignore_c0);


/* ---- converter fit: EM building blocks --------------------------------------------------------
 * sklearn.mixture.GaussianMixture(covariance_type='full').fit as used at
 *                                                    kwiiyatta/converter/gmm.py:14-26
 * The data are sharded by frames; each call works on the local shard X (n x D, device).
 * A driver (kwiiyatta_amd/converter/gmm_fit.py) runs, per EM iteration,
 *   estep -> sums -> [all-reduce stats] -> means -> cov -> [all-reduce sxx] -> finalize
 * and all-reduces the statistics over RCCL when several GPUs hold shards.
 * M <= 256, D <= 160 (and (D+1) / (256 / 2^ceil(log2 M)) <= 40). */
int kwy_gmm_em_estep_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *weights,
                         const double *means, const double *covs, double *resp /* n x M */,
                         double *loglik_parts /* ceil(n/256) */, int *status_out);
int kwy_gmm_em_sums_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                        double *stats /* M x (D+1): [sum r, sum r x] */);
int kwy_gmm_em_means_dev(kwy_ctx *ctx, const double *stats, int D, int M, double *means);
int kwy_gmm_em_cov_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                       const double *means, double *sxx /* M x D x D */);
/* the same with the (globally reduced) sums of kwy_gmm_em_sums_dev at hand (stats; NULL = kwy_gmm_em_cov_dev): a
 * frame whose responsibility for a mixture is below 2^-70 of that mixture's mass nk is left out of its covariance
 * sum -- all such frames together weigh less than n 2^-70 of the mixture, below the rounding of nk itself for
 * n <= 2^25 -- which lets the kernel skip the matrix products of the (many) frames that do not belong to a
 * component.  kwy_gmm_em_cov_dev skips below 2^-200 only. */
int kwy_gmm_em_cov_stats_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                             const double *means, const double *stats, double *sxx);
int kwy_gmm_em_finalize_dev(kwy_ctx *ctx, const double *stats, const double *sxx, int D, int M,
                            double reg_covar, double *weights, double *covs);
int kwy_gmm_em_scratch_bytes(int64_t n, int D, int M, int64_t *bytes);

/* The whole fit of ONE rank's rows as one call: GaussianMixture(n_components=M, covariance_type='full', max_iter, tol,
 * reg_covar, random_state=seed).fit(X)                kwiiyatta/converter/gmm.py:14-26
 * -- the k-means initialisation below (numpy's RandomState(seed) draws, scikit-learn's seeding and Lloyd loop) and the
 * EM loop above, run by the library.  X: n x D on the device; weights (M), means (M x D), covs (M x D x D): HOST outputs;
 * n_iter / lower_bound / converged / kmeans_iter: scikit-learn's n_iter_, lower_bound_, converged_ and the number of
 * Lloyd iterations (may be NULL).  Synchronous. */
int kwy_gmm_fit_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int max_iter, double tol, double reg_covar,
                    uint32_t seed, double *weights, double *means, double *covs, int *n_iter, double *lower_bound,
                    int *converged, int *kmeans_iter);
/* The same fit over the rows of SEVERAL ranks (fit row: "multi-GPU variant takes an RCCL communicator
 * handle"): every rank calls with its own shard X (global row order: rank 0's rows, then rank 1's, ...) and a
 * communicator given as an in-place SUM all-reduce of device doubles on a HIP stream --
 *     int reduce(void *user, double *buf, int64_t count, void *stream) {
 *       return ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)user, (hipStream_t)stream) != ncclSuccess; }
 * -- and all ranks return the same model (that of a one-rank fit of the concatenated rows, up to the rounding of the
 * sums).  What is exchanged: per k-means++ centre the shard totals, <= 8 candidate rows and potentials; per Lloyd
 * iteration M (D + 1) + 1 doubles; per EM iteration M (D + 1) and M D D doubles and the log-likelihood.  comm == NULL:
 * kwy_gmm_fit_dev.  The library does not link RCCL: the callback is the caller's.
 * Errors: what can go wrong on ONE rank -- its arguments (every rank needs n >= 1 rows: an empty shard is an error),
 * its allocations -- is the first quantity the ranks exchange, so either every rank fits or every rank returns an
 * error (the failing ones their own code, the others KWY_EINVAL "another rank failed its set-up"); nobody is left
 * waiting in a collective.  A failure of the callback itself (non-zero return, e.g. an exception in a Python
 * binder) cannot be agreed on and is fatal for the whole group: the caller must tear the group down. */
typedef struct kwy_comm {
  int rank, world;
  int (*all_reduce_sum)(void *user, double *device_buffer, int64_t count, void *stream);   /* 0 = success */
  void *user;
} kwy_comm;
int kwy_gmm_fit_comm_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int max_iter, double tol,
                         double reg_covar, uint32_t seed, const kwy_comm *comm, double *weights, double *means,
                         double *covs, int *n_iter, double *lower_bound, int *converged, int *kmeans_iter);

/* ---- converter fit: k-means initialisation ------------------------------------------------------
 * The `init_params='kmeans'` step of the same GaussianMixture.fit (kwiiyatta/converter/gmm.py:14-23):
 * sklearn.cluster.KMeans(n_clusters=M, n_init=1) -- centred input, k-means++ seeding with 2 + ln M
 * candidates per centre, Lloyd iterations until the labels stop changing or the summed squared centre
 * shift falls below 1e-4 mean(var(X)) -- whose labels become the one-hot responsibilities of the first
 * M-step.  Sharded by rows like the EM blocks; the driver all-reduces / all-gathers shard totals,
 * candidate rows, potentials and the centroid statistics ("k-means init likewise").
 * All pointers are device pointers; D <= 160, M <= 256, at most 8 candidates per step. */
/* out[0..D) = sum_t (X[t] - shift), out[D..2D) = sum_t (X[t] - shift)^2; shift may be NULL */
int kwy_km_colstats_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *shift, double *out);
/* Xc = X - mean, xsq[t] = |Xc[t]|^2 */
int kwy_km_center_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *mean, double *Xc,
                      double *xsq);
/* k-means++: newd[c][t] = min(closest[t], max(0, |Xc[t] - cand[c]|^2)) for c < L (closest NULL: no min),
 * pots[c] = sum_t newd[c][t] over this shard.  newd: L x n, pots: 8 doubles. */
int kwy_km_pp_dist_dev(kwy_ctx *ctx, const double *Xc, const double *xsq, int64_t n, int D, const double *cand,
                       int L, const double *closest, double *newd, double *pots);
/* np.searchsorted(np.cumsum(closest), vals) over the global row order, shard by shard:
 * kwy_km_pp_total_dev fills csums (kwy_km_chunks(n) chunk totals) and total[0]; with lo / hi = the cumulated totals
 * of the shards before this one / including this one (the same cumulative sums of the all-gathered totals on every
 * rank, so that a value on a shard boundary has exactly one owner; NULL: 0 / lo + this shard's total),
 * kwy_km_pp_pick_dev returns for each vals[c] the local row index of the hit, or -1 if it lies in another shard
 * (first / last: position of this shard; beyond the end -> last row). */
int64_t kwy_km_chunks(int64_t n);
int kwy_km_pp_total_dev(kwy_ctx *ctx, const double *v, int64_t n, double *csums, double *total);
int kwy_km_pp_pick_dev(kwy_ctx *ctx, const double *v, int64_t n, const double *csums, const double *lo,
                       const double *hi, const double *vals, int L, int first, int last, int64_t *idx);
/* Lloyd: labels[t] = argmin_j |c_j|^2 - 2 <Xc[t], c_j> (int32; in: previous labels), resp = one-hot rows
 * (n x M, may be NULL), changed[0] = number of rows whose label changed (uint64).  The centroid sums are
 * kwy_gmm_em_sums_dev(Xc, resp); kwy_km_update_dev turns the reduced [count, sums] into the new centres
 * and the squared shift of every centre. */
int kwy_km_assign_dev(kwy_ctx *ctx, const double *Xc, int64_t n, int D, const double *centers, int M,
                      int32_t *labels, double *resp, unsigned long long *changed);
int kwy_km_update_dev(kwy_ctx *ctx, const double *stats, const double *centers_old, int M, int D,
                      double *centers_new, double *shift2);
/* Up to `iterations` Lloyd iterations back to back WITHOUT the host (one shard: no reduction between the ranks' sums):
 * each is kwy_km_assign_dev + kwy_gmm_em_sums_dev + kwy_km_update_dev followed by the decision sklearn's
 * _kmeans_single_lloyd takes after an iteration, taken on the device; the kernels of the iterations enqueued behind
 * the one that ended the loop return at once.  centers2: 2 x M x D, the centres of iteration i in buffer i & 1.
 * state (int64[4], zeroed by the caller before the first batch): [0] 0 = running, 1 = labels unchanged (strict
 * convergence), 2 = squared centre shift <= abs_tol (summed in numpy's order), 3 = a cluster is empty -- the
 * iteration is left after its sums (labels, resp, stats are its own, the centres not updated) for the caller to
 * relocate and finish (resp is NOT up to date: kwy_km_onehot_dev), 4 = max_iter iterations done; [1] finished iterations; [2] changed labels of the last
 * assignment.  log: 2 doubles per finished iteration (changed labels, centre shift), max_iter rows.
 * Replaces the per-iteration host round trip of sklearn/cluster/_kmeans.py:_kmeans_single_lloyd. */
int kwy_km_lloyd_dev(kwy_ctx *ctx, const double *Xc, int64_t n, int D, double *centers2, int M, int32_t *labels,
                     double *resp, unsigned long long *changed, double *stats, double *shift2, double abs_tol,
                     int iterations, int64_t max_iter, long long *state, double *log);
/* resp[t][:] = the one-hot row of labels[t] (n x M): kwy_km_lloyd_dev sums the centroids from the labels and leaves
 * `resp` alone (4 bytes per frame and iteration instead of M doubles written and read); the caller that wants
 * scikit-learn's `resp` of the k-means initialisation (sklearn/mixture/_base.py:_initialize_parameters) asks for it
 * once, after the loop. */
int kwy_km_onehot_dev(kwy_ctx *ctx, const int32_t *labels, int64_t n, int M, double *resp);

/* ---- training-set path (device-resident) ---------------------------------------------------------
 * What the reference does per parallel pair before the converter fit (Config.load_dataset ->
 * align_dataset -> MelCepstrumDataset -> DeltaFeatureDataset -> make_dataset_to_array,
 * kwiiyatta/config.py:83-104, converter/__init__.py:17-18, converter/dataset.py:49-77), as device
 * calls so that the joint feature rows are produced in HBM.  Counts that depend on the data
 * (path length, kept rows) are device scalars; buffers are sized by their capacities. */
/* TrimmedDataset (dataset.py:49-52): n_out[0] = len(trim_zeros_frames(sp)), rows with L1 norm < eps
 * (nnmnkwii: 1e-7) counting as zero; the caller keeps the FIRST n_out frames, as the reference does */
int kwy_trim_length_dev(kwy_ctx *ctx, const double *sp, int64_t T, int K, double eps, int64_t *n_out);
typedef struct kwy_trim_job {
  const double *sp;      /* T x K */
  int64_t T;
  int64_t *n_out;        /* 1 */
} kwy_trim_job;
int kwy_trim_length_batch_dev(kwy_ctx *ctx, const kwy_trim_job *jobs, int count, int K, double eps);
/* pad_silence's cheap parts (kwiiyatta/vocoder/feature.py:19-41) on the first n frames of analysed utterances that are
 * stored with pad_len frames of room on both ends -- f0_pad (n + 2 pad_len): the f0 track between zeros; ap_pad
 * ((n + 2 pad_len) x K): the rows behind the kept frames = 1 - 1e-12 (the rows in front are the caller's
 * initialisation) -- and, when `voiced` is not NULL, WorldSynthesizer.extract_is_voiced of the padded feature
 * (world.py:147-151).  The pad SPECTRA come from kwy_np_normal_blocks_dev. */
typedef struct kwy_pad_job {
  const double *f0;      /* >= n */
  int64_t n;             /* frames kept (kwy_trim_length_dev) */
  double *f0_pad;        /* n + 2 pad_len */
  double *ap_pad;        /* (n + 2 pad_len) x K, rows [pad_len, pad_len + n) analysed */
  double *voiced;        /* n + 2 pad_len, or NULL */
} kwy_pad_job;
int kwy_train_pad_batch_dev(kwy_ctx *ctx, const kwy_pad_job *jobs, int count, int K, int fs, int pad_len);
/* WorldSynthesizer.extract_is_voiced (world.py:147-151): voiced[t] = 1.0 / 0.0 */
int kwy_is_voiced_dev(kwy_ctx *ctx, const double *f0, const double *ap, int64_t T, int K, int fs,
                      double *voiced);
/* dtw_feature's strict filter (align.py:73-92) and align_even's cut to [pad_len, T - pad_len)
 * (align.py:134-146) on a FastDTW path over the DTW features feat_x (Tx x width) / feat_y:
 * idx_x / idx_y <- the x and y of the surviving cells, n_out[0] their number (<= capacity). */
int kwy_align_even_dev(kwy_ctx *ctx, const int32_t *path, const int64_t *path_len, const double *feat_x,
                       const double *feat_y, int width, int strict, int use_power, int use_vuv, int64_t Tx,
                       int64_t Ty, int pad_len, int32_t *idx_x, int32_t *idx_y, int64_t capacity,
                       int64_t *n_out);
/* nnmnkwii delta_features(x, DELTA_WINDOWS) (delta.py:8-12,30): x: n[0] x d (device count, <= capacity
 * rows allocated) -> out: n x 3d = [static | delta | delta-delta], zero-padded at both ends */
int kwy_delta_features_dev(kwy_ctx *ctx, const double *x, const int64_t *n, int64_t capacity, int d,
                           double *out);
/* make_dataset_to_array's np.hstack + remove_zeros_frames (dataset.py:61-77): joint <- [xd[t] | yd[t]]
 * for the rows t < n[0] with L1 norm >= eps, in order; n_out[0] = rows written */
int kwy_joint_rows_dev(kwy_ctx *ctx, const double *xd, const double *yd, const int64_t *n, int64_t capacity,
                       int width, double eps, double *joint, int64_t *n_out);
/* Everything between FastDTW and the training matrix for a batch of aligned pairs, counts staying on the device:
 * dtw_feature's strict filter + align_even's cut (kwy_align_even_dev), the mel-cepstra of the surviving cells without
 * c0, their delta features, np.hstack + remove_zeros_frames, and the APPEND to the matrix
 *   (kwiiyatta/vocoder/align.py:73-92,134-146, converter/mcep.py:10-33, delta.py:15-30, dataset.py:61-77)
 * joint: capacity_rows x (2 * 3 d) doubles; cursor[0]: rows written so far (device; read and advanced in pair order,
 * so the matrix holds the pairs' rows in job order behind whatever was there).  n_rows[0] <- the pair's rows (a pair
 * that does not fit the capacity is dropped whole: n_rows = -1 - rows).  No host synchronisation. */
typedef struct kwy_train_job {
  const int32_t *path;       /* FastDTW path over the padded DTW features */
  const int64_t *path_len;
  const double *feat_x;      /* x_length x (d + 2) DTW features */
  const double *feat_y;      /* y_length x (d + 2) */
  const double *mc_x;        /* x_length x (d + 1) padded mel-cepstra */
  const double *mc_y;        /* y_length x (d + 1) */
  int64_t x_length, y_length;
  int64_t *n_rows;           /* 1 */
} kwy_train_job;
int kwy_train_rows_batch_dev(kwy_ctx *ctx, const kwy_train_job *jobs, int count, int d, int strict, int use_power,
                             int use_vuv, int pad_len, double eps, double *joint, int64_t capacity_rows,
                             int64_t *cursor);

#ifdef __cplusplus
}
#endif
#endif /* KWY_H_ */
