/*
 * oracle/ko_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU oracle (liboracle.so): a plain-C, double-precision,
 * single-threaded restatement of the third-party arithmetic on kwiiyatta's
 * per-utterance conversion path.  It is the parity checker for the HIP
 * product and the "port" CPU baseline of bench.py -- never the product.
 * See the header of each ko_*.c for the reference call sites restated.
 */
#ifndef KO_ORACLE_H_
#define KO_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_MAX_WIN 8 /* max delta-window length */

/* WORLD (ko_world.c) */
int ko_cheaptrick_fft_size(int fs, double f0_floor);
double ko_cheaptrick_f0_floor(int fs, int fft_size);
int ko_d4c_fft_size(int fs);
int ko_d4c_lovetrain_fft_size(int fs);
int ko_d4c_num_bands(int fs);
int64_t ko_dio_samples(int fs, int64_t x_length, double frame_period_ms);
void ko_randn_fill(double *out, int64_t n);

int ko_dio(const double *x, int64_t x_length, int fs, double f0_floor, double f0_ceil,
           double channels_in_octave, double frame_period_ms, int speed,
           double allowed_range, double *temporal_positions, double *f0);
int ko_stonemask(const double *x, int64_t x_length, int fs, const double *t,
                 const double *f0, int64_t f0_length, double *refined_f0);
int ko_cheaptrick(const double *x, int64_t x_length, int fs, const double *t,
                  const double *f0, int64_t f0_length, double q1, double f0_floor,
                  int fft_size, double *out);
int ko_d4c(const double *x, int64_t x_length, int fs, const double *t, const double *f0,
           int64_t f0_length, double threshold, int fft_size, double *out);
int64_t ko_synth_timebase(const double *f0, int64_t f0_length, int fs,
                          double frame_period_ms, int64_t y_length, int fft_size,
                          int32_t *pulse_index, double *pulse_time_shift,
                          double *interpolated_vuv);
int ko_synthesize(const double *f0, int64_t f0_length, const double *spectrogram,
                  const double *aperiodicity, int fft_size, double frame_period_ms,
                  int fs, int64_t y_length, double *y);

int ko_code_aperiodicity(const double *aperiodicity, int64_t f0_length, int fs, int fft_size,
                         double *coded);
int ko_decode_aperiodicity(const double *coded, int64_t f0_length, int fs, int fft_size, int nb,
                           double *aperiodicity);

/* SPTK (ko_sptk.c, ko_mlsa.c) */
void ko_mc2b(const double *mc, int64_t T, int m, double a, double *b);
int ko_mlsadf_delay_length(int m, int pd);
double ko_mlsadf(double x, const double *b, int m, double a, int pd, double *d);
int ko_mlsa_synthesis(const double *x, int64_t n, const double *b, int64_t T, int m, double a, int pd,
                      int hop, double *y);
void ko_freqt(const double *c1, int m1, double *c2, int m2, double a);
int ko_sp2mc(const double *sp, int64_t T, int K, int order, double alpha, double *mc);
int ko_mc2sp(const double *mc, int64_t T, int order, double alpha, int fftlen, double *sp);

/* fastdtw (ko_dtw.c) */
int ko_fastdtw(const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
               int radius, double *dist, int32_t *path, int64_t *path_len);

/* nnmnkwii delta / MLPG (ko_mlpg.c) */
int ko_delta_features(const double *x, int64_t T, int d, int nwin, const int *wl,
                      const int *wu, const double *wcoef, double *out);
int ko_mlpg(const double *mean_frames, const double *variance_frames, int64_t T,
            int static_dim, int nwin, const int *wl, const int *wu,
            const double *wcoef, double *y);
int ko_gmm_mlpg(const double *x, int64_t T, int d, int M, const double *weights,
                const double *means, const double *covs, int diff, int nwin,
                const int *wl, const int *wu, const double *wcoef, double *y,
                int32_t *mix_out);

#ifdef __cplusplus
}
#endif
#endif
