/*
 * oracle/ko_world.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Double-precision, single-threaded C restatement of the WORLD vocoder
 * routines the reference reaches through pyworld 0.2.8 (Pipfile.lock:154):
 *
 *   pyworld.dio / stonemask      kwiiyatta/vocoder/world.py:35-40
 *   pyworld.cheaptrick           kwiiyatta/vocoder/world.py:45
 *   pyworld.d4c                  kwiiyatta/vocoder/world.py:55
 *   pyworld.synthesize           kwiiyatta/vocoder/world.py:86-92
 *   pyworld.get_cheaptrick_fft_size  kwiiyatta/vocoder/world.py:96
 *
 * pyworld's source is NOT vendored under /root/reference and is not installed
 * in this image, so this file restates the published WORLD algorithms
 * (Morise 2015 "CheapTrick", Morise 2016 "D4C", Morise et al. 2016 "WORLD",
 * DIO / StoneMask) as shipped with pyworld 0.2.8.  It is pinned only by the
 * reference's own statistical known-answer envelopes (tests/test_oracle_kat.py,
 * from /root/reference/tests/kwiiyatta/test_vocoder.py:140-181 etc.):
 * sample-level parity with upstream pyworld is UNPINNED.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this file.  The product (kwiiyatta_amd/) never links or calls it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ko_fft.h"
#include "ko_oracle.h"

/* ---- WORLD constants (constantnumbers.h) -------------------------------- */
#define kPi 3.1415926535897932384
#define kMySafeGuardMinimum 0.000000000001
#define kEps 0.00000000000000022204460492503131
#define kFloorF0 71.0
#define kCeilF0 800.0
#define kDefaultF0 500.0
#define kLog2 0.69314718055994529
#define kMaximumValue 100000.0
#define kCutOff 50.0
#define kFloorF0StoneMask 40.0
#define kFrequencyInterval 3000.0
#define kUpperLimit 15000.0
#define kThreshold 0.85
#define kFloorF0D4C 47.0
enum { kHanning = 1, kBlackman = 2 };

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* matlab_round: round half away from zero (WORLD matlabfunctions.cpp). */
static inline int matlab_round(double x) {
  return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5);
}

static double *dalloc(size_t n) { return (double *)malloc(sizeof(double) * (n ? n : 1)); }

/* ---- xorshift128 Gaussian-ish generator (WORLD matlabfunctions.cpp) ------ */
typedef struct { uint32_t x, y, z, w; } ko_rng;

void ko_rng_seed(ko_rng *r) {
  r->x = 123456789u; r->y = 362436069u; r->z = 521288629u; r->w = 88675123u;
}

static inline uint32_t rng_step(ko_rng *r) {
  uint32_t t = r->x ^ (r->x << 11);
  r->x = r->y; r->y = r->z; r->z = r->w;
  r->w = (r->w ^ (r->w >> 19)) ^ (t ^ (t >> 8));
  return r->w;
}

/* sum of 12 uniform draws (w >> 4), scaled to ~N(0,1) */
static inline double rng_randn(ko_rng *r) {
  uint32_t tmp = rng_step(r) >> 4;
  for (int i = 0; i < 11; ++i) tmp += rng_step(r) >> 4;
  return tmp / 268435456.0 - 6.0;
}

void ko_randn_fill(double *out, int64_t n) {
  ko_rng r; ko_rng_seed(&r);
  for (int64_t i = 0; i < n; ++i) out[i] = rng_randn(&r);
}

/* ---- matlabfunctions: histc / interp1 / interp1Q / diff ------------------ */
static void histc(const double *x, int x_length, const double *edges,
                  int edges_length, int *index) {
  int count = 1;
  int i = 0;
  for (; i < edges_length; ++i) {
    index[i] = 1;
    if (edges[i] >= x[0]) break;
  }
  for (; i < edges_length; ++i) {
    if (edges[i] < x[count]) {
      index[i] = count;
    } else {
      index[i--] = count++;
    }
    if (count == x_length) break;
  }
  count--;
  for (i++; i < edges_length; ++i) index[i] = count;
}

static void interp1(const double *x, const double *y, int x_length,
                    const double *xi, int xi_length, double *yi) {
  double *h = dalloc(x_length - 1);
  int *k = (int *)calloc(xi_length ? xi_length : 1, sizeof(int));
  for (int i = 0; i < x_length - 1; ++i) h[i] = x[i + 1] - x[i];
  histc(x, x_length, xi, xi_length, k);
  for (int i = 0; i < xi_length; ++i) {
    double s = (xi[i] - x[k[i] - 1]) / h[k[i] - 1];
    yi[i] = y[k[i] - 1] + s * (y[k[i]] - y[k[i] - 1]);
  }
  free(k);
  free(h);
}

/* interp1 on an equally spaced axis x0 + shift*j (WORLD interp1Q). */
static void interp1Q(double x, double shift, const double *y, int x_length,
                     const double *xi, int xi_length, double *yi) {
  double *delta_y = dalloc(x_length);
  for (int i = 0; i < x_length - 1; ++i) delta_y[i] = y[i + 1] - y[i];
  delta_y[x_length - 1] = 0.0;
  for (int i = 0; i < xi_length; ++i) {
    int base = (int)((xi[i] - x) / shift);
    double frac = (xi[i] - x) / shift - base;
    yi[i] = y[base] + delta_y[base] * frac;
  }
  free(delta_y);
}

/* ---- common.cpp --------------------------------------------------------- */
static int GetSuitableFFTSize(int sample) {
  return (int)pow(2.0, (int)(log((double)sample) / kLog2) + 1.0);
}

static void NuttallWindow(int y_length, double *y) {
  for (int i = 0; i < y_length; ++i) {
    double tmp = i / (y_length - 1.0);
    y[i] = 0.355768 - 0.487396 * cos(2.0 * kPi * tmp) +
           0.144232 * cos(4.0 * kPi * tmp) - 0.012604 * cos(6.0 * kPi * tmp);
  }
}

static void DCCorrection(const double *input, double f0, int fs, int fft_size,
                         double *output) {
  int upper_limit = 2 + (int)(f0 * fft_size / fs);
  double *low_frequency_replica = dalloc(upper_limit);
  double *low_frequency_axis = dalloc(upper_limit);
  for (int i = 0; i < upper_limit; ++i)
    low_frequency_axis[i] = (double)i * fs / fft_size;
  int upper_limit_replica = upper_limit - 1;
  interp1Q(f0 - low_frequency_axis[0], -(double)fs / fft_size, input,
           upper_limit + 1, low_frequency_axis, upper_limit_replica,
           low_frequency_replica);
  for (int i = 0; i < upper_limit_replica; ++i)
    output[i] = input[i] + low_frequency_replica[i];
  free(low_frequency_replica);
  free(low_frequency_axis);
}

static void LinearSmoothing(const double *input, double width, int fs,
                            int fft_size, double *output) {
  int boundary = (int)(width * fft_size / fs) + 1;
  int half = fft_size / 2;
  int mlen = half + boundary * 2 + 1;
  double *mirroring_spectrum = dalloc(mlen);
  double *mirroring_segment = dalloc(mlen);
  double *frequency_axis = dalloc(half + 1);
  for (int i = 0; i < boundary; ++i)
    mirroring_spectrum[i] = input[boundary - i];
  for (int i = boundary; i < half + boundary; ++i)
    mirroring_spectrum[i] = input[i - boundary];
  for (int i = half + boundary; i <= half + boundary * 2; ++i)
    mirroring_spectrum[i] = input[half - (i - (half + boundary))];

  mirroring_segment[0] = mirroring_spectrum[0] * fs / fft_size;
  for (int i = 1; i < mlen; ++i)
    mirroring_segment[i] = mirroring_spectrum[i] * fs / fft_size + mirroring_segment[i - 1];

  for (int i = 0; i <= half; ++i)
    frequency_axis[i] = (double)i / fft_size * fs - width / 2.0;

  double *low_levels = dalloc(half + 1);
  double *high_levels = dalloc(half + 1);
  double origin_of_mirroring_axis = -(boundary - 0.5) * fs / fft_size;
  double discrete_frequency_interval = (double)fs / fft_size;

  interp1Q(origin_of_mirroring_axis, discrete_frequency_interval,
           mirroring_segment, mlen, frequency_axis, half + 1, low_levels);
  for (int i = 0; i <= half; ++i) frequency_axis[i] += width;
  interp1Q(origin_of_mirroring_axis, discrete_frequency_interval,
           mirroring_segment, mlen, frequency_axis, half + 1, high_levels);

  for (int i = 0; i <= half; ++i)
    output[i] = (high_levels[i] - low_levels[i]) / width;

  free(mirroring_spectrum); free(mirroring_segment); free(frequency_axis);
  free(low_levels); free(high_levels);
}

/* Minimum-phase spectrum of a log-amplitude half spectrum (common.cpp
 * GetMinimumPhaseSpectrum).  log_spectrum: fft/2+1 values in, scratch of fft;
 * out: fft/2+1 interleaved complex. */
static void GetMinimumPhaseSpectrum(double *log_spectrum, int fft_size,
                                    double *cep /* 2*fft */, double *out) {
  int half = fft_size / 2;
  for (int i = half + 1; i < fft_size; ++i)
    log_spectrum[i] = log_spectrum[fft_size - i];
  /* WORLD runs a forward r2c here and flips the sign of the imaginary part */
  ko_rfft(log_spectrum, fft_size, cep);
  cep[1] *= -1.0;
  for (int i = 1; i < half; ++i) {
    cep[2 * i] *= 2.0;
    cep[2 * i + 1] *= -2.0;
  }
  cep[2 * half + 1] *= -1.0;
  for (int i = half + 1; i < fft_size; ++i) {
    cep[2 * i] = 0.0;
    cep[2 * i + 1] = 0.0;
  }
  ko_cfft(cep, fft_size, -1);
  for (int i = 0; i <= half; ++i) {
    double tmp = exp(cep[2 * i] / fft_size);
    double ph = cep[2 * i + 1] / fft_size;
    out[2 * i] = tmp * cos(ph);
    out[2 * i + 1] = tmp * sin(ph);
  }
}

/* ---- CheapTrick ---------------------------------------------------------- */
int ko_cheaptrick_fft_size(int fs, double f0_floor) {
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / f0_floor + 1) / kLog2));
}

double ko_cheaptrick_f0_floor(int fs, int fft_size) {
  return 3.0 * fs / (fft_size - 3.0);
}

static void ct_windowed_waveform(const double *x, int x_length, int fs,
                                 double current_f0, double current_position,
                                 ko_rng *rng, double *waveform) {
  int half_window_length = matlab_round(1.5 * fs / current_f0);
  int wl = half_window_length * 2 + 1;
  double *window = dalloc(wl);
  int origin = matlab_round(current_position * fs + 0.001);
  double average = 0.0;
  for (int i = 0; i < wl; ++i) {
    double position = (i - half_window_length) / 1.5 / fs;
    window[i] = 0.5 * cos(kPi * position * current_f0) + 0.5;
    average += window[i] * window[i];
  }
  average = sqrt(average);
  for (int i = 0; i < wl; ++i) window[i] /= average;

  for (int i = 0; i < wl; ++i) {
    int safe = imin(x_length - 1, imax(0, origin + i - half_window_length));
    waveform[i] = x[safe] * window[i] + rng_randn(rng) * 0.000000000000001;
  }
  double tmp_weight1 = 0, tmp_weight2 = 0;
  for (int i = 0; i < wl; ++i) {
    tmp_weight1 += waveform[i];
    tmp_weight2 += window[i];
  }
  double weighting_coefficient = tmp_weight1 / tmp_weight2;
  for (int i = 0; i < wl; ++i) waveform[i] -= window[i] * weighting_coefficient;
  free(window);
}

static void cheaptrick_frame(const double *x, int x_length, int fs,
                             double current_f0, int fft_size,
                             double current_position, double q1, ko_rng *rng,
                             double *waveform /* fft */, double *spec /* 2*(fft/2+1) */,
                             double *spectral_envelope) {
  int half = fft_size / 2;
  int half_window_length = matlab_round(1.5 * fs / current_f0);
  /* F0-adaptive windowing */
  ct_windowed_waveform(x, x_length, fs, current_f0, current_position, rng, waveform);
  /* power spectrum + DC correction */
  for (int i = half_window_length * 2 + 1; i < fft_size; ++i) waveform[i] = 0.0;
  ko_rfft(waveform, fft_size, spec);
  double *power_spectrum = waveform;
  for (int i = 0; i <= half; ++i)
    power_spectrum[i] = spec[2 * i] * spec[2 * i] + spec[2 * i + 1] * spec[2 * i + 1];
  DCCorrection(power_spectrum, current_f0, fs, fft_size, power_spectrum);
  /* linear-axis smoothing */
  LinearSmoothing(power_spectrum, current_f0 * 2.0 / 3.0, fs, fft_size, power_spectrum);
  /* infinitesimal noise */
  for (int i = 0; i <= half; ++i)
    power_spectrum[i] = power_spectrum[i] + fabs(rng_randn(rng)) * kEps;
  /* smoothing + spectral recovery in the cepstrum domain */
  for (int i = 0; i <= half; ++i) waveform[i] = log(waveform[i]);
  for (int i = 1; i < half; ++i) waveform[fft_size - i] = waveform[i];
  ko_rfft(waveform, fft_size, spec);
  for (int i = 0; i <= half; ++i) {
    double smoothing_lifter, compensation_lifter;
    if (i == 0) {
      smoothing_lifter = 1.0;
      compensation_lifter = (1.0 - 2.0 * q1) + 2.0 * q1;
    } else {
      double quefrency = (double)i / fs;
      smoothing_lifter = sin(kPi * current_f0 * quefrency) / (kPi * current_f0 * quefrency);
      compensation_lifter = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * kPi * quefrency * current_f0);
    }
    spec[2 * i] = spec[2 * i] * smoothing_lifter * compensation_lifter / fft_size;
    spec[2 * i + 1] = 0.0;
  }
  ko_irfft(spec, fft_size, waveform);
  for (int i = 0; i <= half; ++i) spectral_envelope[i] = exp(waveform[i]);
}

int ko_cheaptrick(const double *x, int64_t x_length, int fs, const double *t,
                  const double *f0, int64_t f0_length, double q1,
                  double f0_floor, int fft_size, double *out) {
  if (fft_size <= 0) fft_size = ko_cheaptrick_fft_size(fs, f0_floor);
  int half = fft_size / 2;
  ko_rng rng; ko_rng_seed(&rng);
  double floor_eff = ko_cheaptrick_f0_floor(fs, fft_size);
  double *waveform = dalloc(fft_size);
  double *spec = dalloc(2 * (half + 1));
  for (int64_t i = 0; i < f0_length; ++i) {
    double current_f0 = f0[i] <= floor_eff ? kDefaultF0 : f0[i];
    cheaptrick_frame(x, (int)x_length, fs, current_f0, fft_size, t[i], q1, &rng,
                     waveform, spec, out + i * (half + 1));
  }
  free(waveform); free(spec);
  return 0;
}

/* ---- D4C ----------------------------------------------------------------- */
static void d4c_windowed_waveform(const double *x, int x_length, int fs,
                                  double current_f0, double current_position,
                                  int window_type, double window_length_ratio,
                                  ko_rng *rng, double *waveform) {
  int half_window_length = matlab_round(window_length_ratio * fs / current_f0 / 2.0);
  int wl = half_window_length * 2 + 1;
  double *window = dalloc(wl);
  int origin = matlab_round(current_position * fs + 0.001);
  for (int i = 0; i < wl; ++i) {
    double position = (2.0 * (i - half_window_length) / window_length_ratio) / fs;
    if (window_type == kHanning)
      window[i] = 0.5 * cos(kPi * position * current_f0) + 0.5;
    else
      window[i] = 0.42 + 0.5 * cos(kPi * position * current_f0) +
                  0.08 * cos(kPi * position * current_f0 * 2);
  }
  for (int i = 0; i < wl; ++i) {
    int safe = imin(x_length - 1, imax(0, origin + i - half_window_length));
    waveform[i] = x[safe] * window[i] + rng_randn(rng) * kMySafeGuardMinimum;
  }
  double tmp_weight1 = 0, tmp_weight2 = 0;
  for (int i = 0; i < wl; ++i) {
    tmp_weight1 += waveform[i];
    tmp_weight2 += window[i];
  }
  double weighting_coefficient = tmp_weight1 / tmp_weight2;
  for (int i = 0; i < wl; ++i) waveform[i] -= window[i] * weighting_coefficient;
  free(window);
}

static void d4c_centroid(const double *x, int x_length, int fs, double current_f0,
                         int fft_size, double current_position, ko_rng *rng,
                         double *waveform, double *spec, double *centroid) {
  int half = fft_size / 2;
  for (int i = 0; i < fft_size; ++i) waveform[i] = 0.0;
  d4c_windowed_waveform(x, x_length, fs, current_f0, current_position, kBlackman,
                        4.0, rng, waveform);
  int wl = matlab_round(2.0 * fs / current_f0) * 2;
  double power = 0.0;
  for (int i = 0; i <= wl; ++i) power += waveform[i] * waveform[i];
  for (int i = 0; i <= wl; ++i) waveform[i] /= sqrt(power);

  ko_rfft(waveform, fft_size, spec);
  double *tmp_real = dalloc(half + 1), *tmp_imag = dalloc(half + 1);
  for (int i = 0; i <= half; ++i) {
    tmp_real[i] = spec[2 * i];
    tmp_imag[i] = spec[2 * i + 1];
  }
  for (int i = 0; i < fft_size; ++i) waveform[i] *= i + 1.0;
  ko_rfft(waveform, fft_size, spec);
  for (int i = 0; i <= half; ++i)
    centroid[i] = spec[2 * i] * tmp_real[i] + tmp_imag[i] * spec[2 * i + 1];
  free(tmp_real); free(tmp_imag);
}

static void d4c_general_body(const double *x, int x_length, int fs,
                             double current_f0, int fft_size,
                             double current_position, int number_of_aperiodicities,
                             const double *window, int window_length,
                             ko_rng *rng, double *waveform, double *spec,
                             double *coarse_aperiodicity) {
  int half = fft_size / 2;
  double *static_centroid = dalloc(half + 1);
  double *smoothed_power_spectrum = dalloc(half + 1);
  double *static_group_delay = dalloc(half + 1);
  double *centroid1 = dalloc(half + 1), *centroid2 = dalloc(half + 1);

  /* static centroid */
  d4c_centroid(x, x_length, fs, current_f0, fft_size,
               current_position - 0.25 / current_f0, rng, waveform, spec, centroid1);
  d4c_centroid(x, x_length, fs, current_f0, fft_size,
               current_position + 0.25 / current_f0, rng, waveform, spec, centroid2);
  for (int i = 0; i <= half; ++i) static_centroid[i] = centroid1[i] + centroid2[i];
  DCCorrection(static_centroid, current_f0, fs, fft_size, static_centroid);

  /* smoothed power spectrum */
  for (int i = 0; i < fft_size; ++i) waveform[i] = 0.0;
  d4c_windowed_waveform(x, x_length, fs, current_f0, current_position, kHanning,
                        4.0, rng, waveform);
  ko_rfft(waveform, fft_size, spec);
  for (int i = 0; i <= half; ++i)
    smoothed_power_spectrum[i] = spec[2 * i] * spec[2 * i] + spec[2 * i + 1] * spec[2 * i + 1];
  DCCorrection(smoothed_power_spectrum, current_f0, fs, fft_size, smoothed_power_spectrum);
  LinearSmoothing(smoothed_power_spectrum, current_f0, fs, fft_size, smoothed_power_spectrum);

  /* static group delay */
  for (int i = 0; i <= half; ++i)
    static_group_delay[i] = static_centroid[i] / smoothed_power_spectrum[i];
  LinearSmoothing(static_group_delay, current_f0 / 2.0, fs, fft_size, static_group_delay);
  double *smoothed_group_delay = dalloc(half + 1);
  LinearSmoothing(static_group_delay, current_f0, fs, fft_size, smoothed_group_delay);
  for (int i = 0; i <= half; ++i) static_group_delay[i] -= smoothed_group_delay[i];

  /* coarse aperiodicity */
  int boundary = matlab_round(fft_size * 8.0 / window_length);
  int half_window_length = window_length / 2;
  for (int i = 0; i < fft_size; ++i) waveform[i] = 0.0;
  double *power_spectrum = dalloc(half + 1);
  for (int i = 0; i < number_of_aperiodicities; ++i) {
    int center = (int)(kFrequencyInterval * (i + 1) * fft_size / fs);
    for (int j = 0; j <= half_window_length * 2; ++j)
      waveform[j] = static_group_delay[center - half_window_length + j] * window[j];
    ko_rfft(waveform, fft_size, spec);
    for (int j = 0; j <= half; ++j)
      power_spectrum[j] = spec[2 * j] * spec[2 * j] + spec[2 * j + 1] * spec[2 * j + 1];
    /* std::sort ascending */
    {
      extern int ko_cmp_double(const void *, const void *);
      qsort(power_spectrum, half + 1, sizeof(double), ko_cmp_double);
    }
    for (int j = 1; j <= half; ++j) power_spectrum[j] += power_spectrum[j - 1];
    coarse_aperiodicity[i] =
        10 * log10(power_spectrum[half - boundary - 1] / power_spectrum[half]);
  }
  /* revision of the result based on the F0 */
  for (int i = 0; i < number_of_aperiodicities; ++i)
    coarse_aperiodicity[i] = dmin(0.0, coarse_aperiodicity[i] + (current_f0 - 100) / 50.0);

  free(static_centroid); free(smoothed_power_spectrum); free(static_group_delay);
  free(centroid1); free(centroid2); free(smoothed_group_delay); free(power_spectrum);
}

int ko_cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int ko_d4c_fft_size(int fs) {
  return (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / kFloorF0D4C + 1) / kLog2));
}

int ko_d4c_lovetrain_fft_size(int fs) {
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / 40.0 + 1) / kLog2));
}

int ko_d4c_num_bands(int fs) {
  return (int)(dmin(kUpperLimit, fs / 2.0 - kFrequencyInterval) / kFrequencyInterval);
}

static double d4c_lovetrain_sub(const double *x, int fs, int x_length,
                                double current_f0, double current_position,
                                int fft_size, int boundary0, int boundary1,
                                int boundary2, ko_rng *rng, double *waveform,
                                double *spec) {
  /* WORLD allocates fft_size doubles and fills bins 0..fft_size/2 only, but for fs < 15.8 kHz the
   * 7.9 kHz boundary lies above fft_size/2 and the cumulative sum runs into uninitialised memory
   * (undefined upstream).  Defined here: the bins above Nyquist are zero. */
  double *power_spectrum = (double *)calloc(fft_size ? fft_size : 1, sizeof(double));
  int window_length = matlab_round(1.5 * fs / current_f0) * 2 + 1;
  d4c_windowed_waveform(x, x_length, fs, current_f0, current_position, kBlackman,
                        3.0, rng, waveform);
  for (int i = window_length; i < fft_size; ++i) waveform[i] = 0.0;
  ko_rfft(waveform, fft_size, spec);
  for (int i = 0; i <= boundary0; ++i) power_spectrum[i] = 0.0;
  for (int i = boundary0 + 1; i < fft_size / 2 + 1; ++i)
    power_spectrum[i] = spec[2 * i] * spec[2 * i] + spec[2 * i + 1] * spec[2 * i + 1];
  for (int i = boundary0; i <= boundary2; ++i)
    power_spectrum[i] += +power_spectrum[i - 1];
  double aperiodicity0 = power_spectrum[boundary1] / power_spectrum[boundary2];
  free(power_spectrum);
  return aperiodicity0;
}

int ko_d4c(const double *x, int64_t x_length_, int fs, const double *t,
           const double *f0, int64_t f0_length, double threshold, int fft_size,
           double *out) {
  int x_length = (int)x_length_;
  if (fft_size <= 0) fft_size = ko_cheaptrick_fft_size(fs, kFloorF0);
  int K = fft_size / 2 + 1;
  ko_rng rng; ko_rng_seed(&rng);
  for (int64_t i = 0; i < f0_length * K; ++i) out[i] = 1.0 - kMySafeGuardMinimum;

  int fft_size_d4c = ko_d4c_fft_size(fs);
  int number_of_aperiodicities = ko_d4c_num_bands(fs);
  int window_length = (int)(kFrequencyInterval * fft_size_d4c / fs) * 2 + 1;
  double *window = dalloc(window_length);
  NuttallWindow(window_length, window);

  /* D4C Love Train: aperiodicity of 0 Hz is given by a different algorithm */
  double *aperiodicity0 = dalloc(f0_length);
  {
    double lowest_f0 = 40.0;
    int fft_l = ko_d4c_lovetrain_fft_size(fs);
    double *waveform = dalloc(fft_l), *spec = dalloc(2 * (fft_l / 2 + 1));
    int boundary0 = (int)ceil(100.0 * fft_l / fs);
    int boundary1 = (int)ceil(4000.0 * fft_l / fs);
    int boundary2 = (int)ceil(7900.0 * fft_l / fs);
    for (int64_t i = 0; i < f0_length; ++i) {
      if (f0[i] == 0.0) { aperiodicity0[i] = 0.0; continue; }
      aperiodicity0[i] = d4c_lovetrain_sub(x, fs, x_length, dmax(f0[i], lowest_f0),
                                           t[i], fft_l, boundary0, boundary1,
                                           boundary2, &rng, waveform, spec);
    }
    free(waveform); free(spec);
  }

  double *coarse_aperiodicity = dalloc(number_of_aperiodicities + 2);
  coarse_aperiodicity[0] = -60.0;
  coarse_aperiodicity[number_of_aperiodicities + 1] = -kMySafeGuardMinimum;
  double *coarse_frequency_axis = dalloc(number_of_aperiodicities + 2);
  for (int i = 0; i <= number_of_aperiodicities; ++i)
    coarse_frequency_axis[i] = i * kFrequencyInterval;
  coarse_frequency_axis[number_of_aperiodicities + 1] = fs / 2.0;
  double *frequency_axis = dalloc(K);
  for (int i = 0; i < K; ++i) frequency_axis[i] = (double)i * fs / fft_size;

  double *waveform = dalloc(fft_size_d4c), *spec = dalloc(2 * (fft_size_d4c / 2 + 1));
  for (int64_t i = 0; i < f0_length; ++i) {
    if (f0[i] == 0 || aperiodicity0[i] <= threshold) continue;
    d4c_general_body(x, x_length, fs, dmax(kFloorF0D4C, f0[i]), fft_size_d4c, t[i],
                     number_of_aperiodicities, window, window_length, &rng,
                     waveform, spec, &coarse_aperiodicity[1]);
    double *ap = out + i * K;
    interp1(coarse_frequency_axis, coarse_aperiodicity, number_of_aperiodicities + 2,
            frequency_axis, K, ap);
    for (int j = 0; j < K; ++j) ap[j] = pow(10.0, ap[j] / 20.0);
  }
  free(waveform); free(spec); free(window); free(aperiodicity0);
  free(coarse_aperiodicity); free(coarse_frequency_axis); free(frequency_axis);
  return 0;
}

/* ---- codec.cpp: aperiodicity band codec ------------------------------------
 * pyworld.code_aperiodicity / decode_aperiodicity, used by the reference only when
 * features move between sampling rates or spectrum lengths
 * (kwiiyatta/vocoder/world.py:98-145).  WORLD codec.cpp as shipped with pyworld 0.2.8:
 * the dB aperiodicity is sampled at 3 kHz, 6 kHz, ... (interp1Q on the bin grid) and
 * rebuilt by linear interpolation over {0: -60 dB, 3k.., fs/2: -1e-12 dB}; a frame whose
 * mean coded value exceeds -0.5 dB is "unvoiced" and decodes to 1 - 1e-12. */
int ko_code_aperiodicity(const double *aperiodicity, int64_t f0_length, int fs, int fft_size,
                         double *coded) {
  const int nb = ko_d4c_num_bands(fs), K = fft_size / 2 + 1;
  if (nb <= 0) return 0;
  double *axis = dalloc(nb), *logap = dalloc(K);
  for (int i = 0; i < nb; ++i) axis[i] = kFrequencyInterval * (i + 1.0);
  for (int64_t i = 0; i < f0_length; ++i) {
    for (int j = 0; j < K; ++j) logap[j] = 20 * log10(aperiodicity[i * K + j]);
    interp1Q(0, (double)fs / fft_size, logap, K, axis, nb, coded + i * nb);
  }
  free(axis); free(logap);
  return 0;
}

int ko_decode_aperiodicity(const double *coded, int64_t f0_length, int fs, int fft_size,
                           int nb, double *aperiodicity) {
  /* nb = number of coded columns = GetNumberOfAperiodicities(fs) in WORLD */
  const int K = fft_size / 2 + 1;
  for (int64_t i = 0; i < f0_length * K; ++i) aperiodicity[i] = 1.0 - kMySafeGuardMinimum;
  if (nb <= 0) return 0;
  double *faxis = dalloc(K), *caxis = dalloc(nb + 2), *cap = dalloc(nb + 2);
  for (int i = 0; i < K; ++i) faxis[i] = (double)fs / fft_size * i;
  for (int i = 0; i <= nb; ++i) caxis[i] = i * kFrequencyInterval;
  caxis[nb + 1] = fs / 2.0;
  cap[0] = -60.0;
  cap[nb + 1] = -kMySafeGuardMinimum;
  for (int64_t i = 0; i < f0_length; ++i) {
    double tmp = 0.0;
    for (int j = 0; j < nb; ++j) { tmp += coded[i * nb + j]; cap[j + 1] = coded[i * nb + j]; }
    tmp /= nb;
    if (tmp > -0.5) continue;
    double *ap = aperiodicity + i * K;
    interp1(caxis, cap, nb + 2, faxis, K, ap);
    for (int j = 0; j < K; ++j) ap[j] = pow(10.0, ap[j] / 20.0);
  }
  free(faxis); free(caxis); free(cap);
  return 0;
}

/* ---- Synthesis ------------------------------------------------------------ */
static inline double GetSafeAperiodicity(double x) {
  return dmax(0.001, dmin(0.999999999999, x));
}

static void GetDCRemover(int fft_size, double *dc_remover) {
  double dc_component = 0.0;
  for (int i = 0; i < fft_size / 2; ++i) {
    dc_remover[i] = 0.5 - 0.5 * cos(2.0 * kPi * (i + 1.0) / (1.0 + fft_size));
    dc_remover[fft_size - i - 1] = dc_remover[i];
    dc_component += dc_remover[i] * 2.0;
  }
  for (int i = 0; i < fft_size / 2; ++i) {
    dc_remover[i] /= dc_component;
    dc_remover[fft_size - i - 1] = dc_remover[i];
  }
}

/* Pulse time base (synthesis.cpp GetTimeBase & friends).  Returns the pulse
 * count; outputs pulse sample index, fractional shift [s] and per-sample vuv. */
int64_t ko_synth_timebase(const double *f0, int64_t f0_length, int fs,
                          double frame_period_ms, int64_t y_length, int fft_size,
                          int32_t *pulse_index, double *pulse_time_shift,
                          double *interpolated_vuv) {
  double frame_period = frame_period_ms / 1000.0;
  double lowest_f0 = fs / fft_size + 1.0; /* integer division, as upstream */
  double *time_axis = dalloc(y_length);
  double *coarse_time_axis = dalloc(f0_length + 1);
  double *coarse_f0 = dalloc(f0_length + 1);
  double *coarse_vuv = dalloc(f0_length + 1);
  for (int64_t i = 0; i < y_length; ++i) time_axis[i] = i / (double)fs;
  for (int64_t i = 0; i < f0_length; ++i) {
    coarse_time_axis[i] = i * frame_period;
    coarse_f0[i] = f0[i] < lowest_f0 ? 0.0 : f0[i];
    coarse_vuv[i] = coarse_f0[i] == 0.0 ? 0.0 : 1.0;
  }
  coarse_time_axis[f0_length] = f0_length * frame_period;
  coarse_f0[f0_length] = coarse_f0[f0_length - 1] * 2 - coarse_f0[f0_length - 2];
  coarse_vuv[f0_length] = coarse_vuv[f0_length - 1] * 2 - coarse_vuv[f0_length - 2];

  double *interpolated_f0 = dalloc(y_length);
  interp1(coarse_time_axis, coarse_f0, (int)f0_length + 1, time_axis, (int)y_length, interpolated_f0);
  interp1(coarse_time_axis, coarse_vuv, (int)f0_length + 1, time_axis, (int)y_length, interpolated_vuv);
  for (int64_t i = 0; i < y_length; ++i) {
    interpolated_vuv[i] = interpolated_vuv[i] > 0.5 ? 1.0 : 0.0;
    interpolated_f0[i] = interpolated_vuv[i] == 0.0 ? kDefaultF0 : interpolated_f0[i];
  }

  double *total_phase = dalloc(y_length);
  double *wrap_phase = dalloc(y_length);
  total_phase[0] = 2.0 * kPi * interpolated_f0[0] / fs;
  wrap_phase[0] = fmod(total_phase[0], 2.0 * kPi);
  for (int64_t i = 1; i < y_length; ++i) {
    total_phase[i] = total_phase[i - 1] + 2.0 * kPi * interpolated_f0[i] / fs;
    wrap_phase[i] = fmod(total_phase[i], 2.0 * kPi);
  }
  int64_t number_of_pulses = 0;
  for (int64_t i = 0; i < y_length - 1; ++i) {
    if (fabs(wrap_phase[i + 1] - wrap_phase[i]) > kPi) {
      pulse_index[number_of_pulses] = (int32_t)i;
      double y1 = wrap_phase[i] - 2.0 * kPi;
      double y2 = wrap_phase[i + 1];
      double xx = -y1 / (y2 - y1);
      pulse_time_shift[number_of_pulses] = xx / fs;
      ++number_of_pulses;
    }
  }
  free(time_axis); free(coarse_time_axis); free(coarse_f0); free(coarse_vuv);
  free(interpolated_f0); free(total_phase); free(wrap_phase);
  return number_of_pulses;
}

int ko_synthesize(const double *f0, int64_t f0_length, const double *spectrogram,
                  const double *aperiodicity, int fft_size, double frame_period_ms,
                  int fs, int64_t y_length, double *y) {
  int half = fft_size / 2, K = half + 1;
  ko_rng rng; ko_rng_seed(&rng);
  for (int64_t i = 0; i < y_length; ++i) y[i] = 0.0;
  if (y_length < 2 || f0_length < 2) return 0;

  int32_t *pulse_index = (int32_t *)malloc(sizeof(int32_t) * y_length);
  double *pulse_shift = dalloc(y_length);
  double *interpolated_vuv = dalloc(y_length);
  int64_t number_of_pulses = ko_synth_timebase(f0, f0_length, fs, frame_period_ms,
                                               y_length, fft_size, pulse_index,
                                               pulse_shift, interpolated_vuv);
  double *dc_remover = dalloc(fft_size);
  GetDCRemover(fft_size, dc_remover);
  double frame_period = frame_period_ms / 1000.0;

  double *spectral_envelope = dalloc(K), *aperiodic_ratio = dalloc(K);
  double *log_spectrum = dalloc(fft_size), *cep = dalloc(2 * fft_size);
  double *mps = dalloc(2 * K), *ispec = dalloc(2 * K), *nspec = dalloc(2 * K);
  double *wave = dalloc(fft_size);
  double *periodic_response = dalloc(fft_size), *aperiodic_response = dalloc(fft_size);

  for (int64_t p = 0; p < number_of_pulses; ++p) {
    int64_t nxt = p + 1 < number_of_pulses ? p + 1 : number_of_pulses - 1;
    int noise_size = pulse_index[nxt] - pulse_index[p];
    double current_vuv = interpolated_vuv[pulse_index[p]];
    double current_time = pulse_index[p] / (double)fs; /* time_axis[i] */

    /* spectral envelope / aperiodic ratio at the pulse time */
    int fl = imin((int)f0_length - 1, (int)floor(current_time / frame_period));
    int ce = imin((int)f0_length - 1, (int)ceil(current_time / frame_period));
    double interpolation = current_time / frame_period - fl;
    const double *s0 = spectrogram + (int64_t)fl * K, *s1 = spectrogram + (int64_t)ce * K;
    const double *a0 = aperiodicity + (int64_t)fl * K, *a1 = aperiodicity + (int64_t)ce * K;
    if (fl == ce) {
      for (int i = 0; i <= half; ++i) {
        spectral_envelope[i] = fabs(s0[i]);
        aperiodic_ratio[i] = pow(GetSafeAperiodicity(a0[i]), 2.0);
      }
    } else {
      for (int i = 0; i <= half; ++i) {
        spectral_envelope[i] = (1.0 - interpolation) * fabs(s0[i]) + interpolation * fabs(s1[i]);
        aperiodic_ratio[i] = pow((1.0 - interpolation) * GetSafeAperiodicity(a0[i]) +
                                 interpolation * GetSafeAperiodicity(a1[i]), 2.0);
      }
    }

    /* periodic response */
    if (current_vuv <= 0.5 || aperiodic_ratio[0] > 0.999) {
      for (int i = 0; i < fft_size; ++i) periodic_response[i] = 0.0;
    } else {
      for (int i = 0; i <= half; ++i)
        log_spectrum[i] = log(spectral_envelope[i] * (1.0 - aperiodic_ratio[i]) +
                              kMySafeGuardMinimum) / 2.0;
      GetMinimumPhaseSpectrum(log_spectrum, fft_size, cep, mps);
      /* fractional time shift by a linear phase (upstream takes the sine as
       * sqrt(1-cos^2), i.e. non-negative -- kept as is) */
      double coefficient = 2.0 * kPi * pulse_shift[p] * fs / fft_size;
      for (int i = 0; i <= half; ++i) {
        double re = mps[2 * i], im = mps[2 * i + 1];
        double re2 = cos(coefficient * i);
        double im2 = sqrt(1.0 - re2 * re2);
        ispec[2 * i] = re * re2 + im * im2;
        ispec[2 * i + 1] = im * re2 - re * im2;
      }
      ko_irfft(ispec, fft_size, wave);
      for (int i = 0; i < half; ++i) { /* fftshift */
        periodic_response[i] = wave[i + half];
        periodic_response[i + half] = wave[i];
      }
      double dc_component = 0.0;
      for (int i = half; i < fft_size; ++i) dc_component += periodic_response[i];
      for (int i = 0; i < half; ++i) periodic_response[i] = -dc_component * dc_remover[i];
      for (int i = half; i < fft_size; ++i) periodic_response[i] -= dc_component * dc_remover[i];
    }

    /* aperiodic response */
    {
      double average = 0.0;
      for (int i = 0; i < noise_size; ++i) {
        wave[i] = rng_randn(&rng);
        average += wave[i];
      }
      average /= noise_size;
      for (int i = 0; i < noise_size; ++i) wave[i] -= average;
      for (int i = noise_size; i < fft_size; ++i) wave[i] = 0.0;
      ko_rfft(wave, fft_size, nspec);
      if (current_vuv != 0.0)
        for (int i = 0; i <= half; ++i)
          log_spectrum[i] = log(spectral_envelope[i] * aperiodic_ratio[i]) / 2.0;
      else
        for (int i = 0; i <= half; ++i) log_spectrum[i] = log(spectral_envelope[i]) / 2.0;
      GetMinimumPhaseSpectrum(log_spectrum, fft_size, cep, mps);
      for (int i = 0; i <= half; ++i) {
        ispec[2 * i] = mps[2 * i] * nspec[2 * i] - mps[2 * i + 1] * nspec[2 * i + 1];
        ispec[2 * i + 1] = mps[2 * i] * nspec[2 * i + 1] + mps[2 * i + 1] * nspec[2 * i];
      }
      ko_irfft(ispec, fft_size, wave);
      for (int i = 0; i < half; ++i) {
        aperiodic_response[i] = wave[i + half];
        aperiodic_response[i + half] = wave[i];
      }
    }

    double sqrt_noise_size = sqrt((double)noise_size);
    int64_t offset = (int64_t)pulse_index[p] - half + 1;
    int lower_limit = (int)(offset < 0 ? -offset : 0);
    int upper_limit = (int)((y_length - offset) < fft_size ? (y_length - offset) : fft_size);
    for (int j = lower_limit; j < upper_limit; ++j)
      y[j + offset] += (periodic_response[j] * sqrt_noise_size + aperiodic_response[j]) / fft_size;
  }

  free(pulse_index); free(pulse_shift); free(interpolated_vuv); free(dc_remover);
  free(spectral_envelope); free(aperiodic_ratio); free(log_spectrum); free(cep);
  free(mps); free(ispec); free(nspec); free(wave);
  free(periodic_response); free(aperiodic_response);
  return 0;
}

/* ---- DIO ------------------------------------------------------------------ */
int64_t ko_dio_samples(int fs, int64_t x_length, double frame_period_ms) {
  return (int64_t)(1000.0 * x_length / fs / frame_period_ms) + 1;
}

static void DesignLowCutFilter(int N, int fft_size, double *low_cut_filter) {
  for (int i = 1; i <= N; ++i)
    low_cut_filter[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * kPi / (N + 1));
  for (int i = N; i < fft_size; ++i) low_cut_filter[i] = 0.0;
  double sum_of_amplitude = 0.0;
  for (int i = 0; i < N; ++i) sum_of_amplitude += low_cut_filter[i];
  for (int i = 0; i < N; ++i) low_cut_filter[i] = -low_cut_filter[i] / sum_of_amplitude;
  for (int i = 0; i < (N - 1) / 2; ++i)
    low_cut_filter[fft_size - (N - 1) / 2 + i] = low_cut_filter[i];
  for (int i = 0; i < N; ++i) low_cut_filter[i] = low_cut_filter[i + (N - 1) / 2];
  low_cut_filter[0] += 1.0;
}

static int ZeroCrossingEngine(const double *filtered_signal, int y_length, double fs,
                              double *interval_locations, double *intervals) {
  int *edges = (int *)malloc(sizeof(int) * (y_length > 0 ? y_length : 1));
  int count = 0;
  for (int i = 0; i < y_length - 1; ++i)
    if (0.0 < filtered_signal[i] && filtered_signal[i + 1] <= 0.0) edges[count++] = i + 1;
  if (count < 2) { free(edges); return 0; }
  double *fine_edges = dalloc(count);
  for (int i = 0; i < count; ++i)
    fine_edges[i] = edges[i] - filtered_signal[edges[i] - 1] /
                    (filtered_signal[edges[i]] - filtered_signal[edges[i] - 1]);
  for (int i = 0; i < count - 1; ++i) {
    intervals[i] = fs / (fine_edges[i + 1] - fine_edges[i]);
    interval_locations[i] = (fine_edges[i] + fine_edges[i + 1]) / 2.0 / fs;
  }
  free(fine_edges); free(edges);
  return count - 1;
}

static void dio_band(double boundary_f0, double fs, const double *y_spectrum,
                     int y_length, int fft_size, double f0_floor, double f0_ceil,
                     const double *temporal_positions, int f0_length,
                     double *f0_score, double *f0_candidate) {
  int half_average_length = matlab_round(fs / boundary_f0 / 2.0);
  double *filtered_signal = dalloc(fft_size);
  {
    double *low_pass_filter = dalloc(fft_size);
    NuttallWindow(half_average_length * 4, low_pass_filter);
    for (int i = half_average_length * 4; i < fft_size; ++i) low_pass_filter[i] = 0.0;
    double *lpf_spec = dalloc(2 * (fft_size / 2 + 1));
    ko_rfft(low_pass_filter, fft_size, lpf_spec);
    for (int i = 0; i <= fft_size / 2; ++i) {
      double tmp = y_spectrum[2 * i] * lpf_spec[2 * i] - y_spectrum[2 * i + 1] * lpf_spec[2 * i + 1];
      lpf_spec[2 * i + 1] = y_spectrum[2 * i] * lpf_spec[2 * i + 1] + y_spectrum[2 * i + 1] * lpf_spec[2 * i];
      lpf_spec[2 * i] = tmp;
    }
    ko_irfft(lpf_spec, fft_size, filtered_signal);
    int index_bias = half_average_length * 2;
    for (int i = 0; i < y_length; ++i) filtered_signal[i] = filtered_signal[i + index_bias];
    free(low_pass_filter); free(lpf_spec);
  }

  double *loc[4], *itv[4];
  int num[4];
  for (int k = 0; k < 4; ++k) { loc[k] = dalloc(y_length); itv[k] = dalloc(y_length); }
  num[0] = ZeroCrossingEngine(filtered_signal, y_length, fs, loc[0], itv[0]);
  for (int i = 0; i < y_length; ++i) filtered_signal[i] = -filtered_signal[i];
  num[1] = ZeroCrossingEngine(filtered_signal, y_length, fs, loc[1], itv[1]);
  for (int i = 0; i < y_length - 1; ++i)
    filtered_signal[i] = filtered_signal[i] - filtered_signal[i + 1];
  num[2] = ZeroCrossingEngine(filtered_signal, y_length - 1, fs, loc[2], itv[2]);
  for (int i = 0; i < y_length - 1; ++i) filtered_signal[i] = -filtered_signal[i];
  num[3] = ZeroCrossingEngine(filtered_signal, y_length - 1, fs, loc[3], itv[3]);

  if (num[0] - 2 <= 0 || num[1] - 2 <= 0 || num[2] - 2 <= 0 || num[3] - 2 <= 0) {
    for (int i = 0; i < f0_length; ++i) {
      f0_score[i] = kMaximumValue;
      f0_candidate[i] = 0.0;
    }
  } else {
    double *set[4];
    for (int k = 0; k < 4; ++k) {
      set[k] = dalloc(f0_length);
      interp1(loc[k], itv[k], num[k], temporal_positions, f0_length, set[k]);
    }
    for (int i = 0; i < f0_length; ++i) {
      double c = (set[0][i] + set[1][i] + set[2][i] + set[3][i]) / 4.0;
      f0_candidate[i] = c;
      f0_score[i] = sqrt(((set[0][i] - c) * (set[0][i] - c) + (set[1][i] - c) * (set[1][i] - c) +
                          (set[2][i] - c) * (set[2][i] - c) + (set[3][i] - c) * (set[3][i] - c)) / 3.0);
      if (c > boundary_f0 || c < boundary_f0 / 2.0 || c > f0_ceil || c < f0_floor) {
        f0_candidate[i] = 0.0;
        f0_score[i] = kMaximumValue;
      }
    }
    for (int k = 0; k < 4; ++k) free(set[k]);
  }
  for (int k = 0; k < 4; ++k) { free(loc[k]); free(itv[k]); }
  free(filtered_signal);
}

static double SelectBestF0(double current_f0, double past_f0, double *const *f0_candidates,
                           int number_of_candidates, int target_index, double allowed_range) {
  double reference_f0 = (current_f0 * 3.0 - past_f0) / 2.0;
  double minimum_error = fabs(reference_f0 - f0_candidates[0][target_index]);
  double best_f0 = f0_candidates[0][target_index];
  for (int i = 1; i < number_of_candidates; ++i) {
    double current_error = fabs(reference_f0 - f0_candidates[i][target_index]);
    if (current_error < minimum_error) {
      minimum_error = current_error;
      best_f0 = f0_candidates[i][target_index];
    }
  }
  if (fabs(1.0 - best_f0 / reference_f0) > allowed_range) return 0.0;
  return best_f0;
}

static void FixF0Contour(double frame_period, int number_of_candidates,
                         double *const *f0_candidates, const double *best_f0_contour,
                         int f0_length, double f0_floor, double allowed_range,
                         double *fixed_f0_contour) {
  int voice_range_minimum = (int)(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1;
  if (f0_length <= voice_range_minimum) return;
  double *f0_tmp1 = dalloc(f0_length), *f0_tmp2 = dalloc(f0_length), *f0_base = dalloc(f0_length);

  /* step 1: prevent jumps */
  for (int i = 0; i < voice_range_minimum; ++i) f0_base[i] = 0.0;
  for (int i = voice_range_minimum; i < f0_length - voice_range_minimum; ++i)
    f0_base[i] = best_f0_contour[i];
  for (int i = f0_length - voice_range_minimum; i < f0_length; ++i) f0_base[i] = 0.0;
  for (int i = 0; i < voice_range_minimum; ++i) f0_tmp1[i] = 0.0;
  for (int i = voice_range_minimum; i < f0_length; ++i)
    f0_tmp1[i] = fabs((f0_base[i] - f0_base[i - 1]) / (kMySafeGuardMinimum + f0_base[i])) <
                 allowed_range ? f0_base[i] : 0.0;

  /* step 2: remove short voiced sections' neighbourhood */
  for (int i = 0; i < f0_length; ++i) f0_tmp2[i] = f0_tmp1[i];
  int center = (voice_range_minimum - 1) / 2;
  for (int i = center; i < f0_length - center; ++i)
    for (int j = -center; j <= center; ++j)
      if (f0_tmp1[i + j] == 0) { f0_tmp2[i] = 0.0; break; }

  int *positive_index = (int *)malloc(sizeof(int) * f0_length);
  int *negative_index = (int *)malloc(sizeof(int) * f0_length);
  int positive_count = 0, negative_count = 0;
  for (int i = 1; i < f0_length; ++i) {
    if (f0_tmp2[i] == 0 && f0_tmp2[i - 1] != 0) negative_index[negative_count++] = i - 1;
    else if (f0_tmp2[i - 1] == 0 && f0_tmp2[i] != 0) positive_index[positive_count++] = i;
  }

  /* step 3: forward extension */
  for (int i = 0; i < f0_length; ++i) f0_tmp1[i] = f0_tmp2[i];
  for (int i = 0; i < negative_count; ++i) {
    int limit = i == negative_count - 1 ? f0_length - 1 : negative_index[i + 1];
    for (int j = negative_index[i]; j < limit; ++j) {
      f0_tmp1[j + 1] = SelectBestF0(f0_tmp1[j], f0_tmp1[j - 1], f0_candidates,
                                    number_of_candidates, j + 1, allowed_range);
      if (f0_tmp1[j + 1] == 0) break;
    }
  }
  /* step 4: backward extension */
  for (int i = 0; i < f0_length; ++i) fixed_f0_contour[i] = f0_tmp1[i];
  for (int i = positive_count - 1; i >= 0; --i) {
    int limit = i == 0 ? 1 : positive_index[i - 1];
    for (int j = positive_index[i]; j > limit; --j) {
      fixed_f0_contour[j - 1] = SelectBestF0(fixed_f0_contour[j], fixed_f0_contour[j + 1],
                                             f0_candidates, number_of_candidates, j - 1,
                                             allowed_range);
      if (fixed_f0_contour[j - 1] == 0) break;
    }
  }
  free(f0_tmp1); free(f0_tmp2); free(f0_base); free(positive_index); free(negative_index);
}

int ko_dio(const double *x, int64_t x_length_, int fs, double f0_floor, double f0_ceil,
           double channels_in_octave, double frame_period, int speed,
           double allowed_range, double *temporal_positions, double *f0) {
  int x_length = (int)x_length_;
  if (speed != 1) return -1; /* decimation path not restated (pyworld default speed=1) */
  int number_of_bands = 1 + (int)(log(f0_ceil / f0_floor) / kLog2 * channels_in_octave);
  double *boundary_f0_list = dalloc(number_of_bands);
  for (int i = 0; i < number_of_bands; ++i)
    boundary_f0_list[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);

  int decimation_ratio = 1;
  int y_length = 1 + (int)(x_length / decimation_ratio);
  double actual_fs = (double)fs / decimation_ratio;
  int fft_size = GetSuitableFFTSize(y_length + matlab_round(actual_fs / kCutOff) * 2 + 1 +
                                    (4 * (int)(1.0 + actual_fs / boundary_f0_list[0] / 2.0)));

  /* spectrum used for the f0 estimation */
  double *y = dalloc(fft_size);
  double *y_spectrum = dalloc(2 * (fft_size / 2 + 1));
  {
    for (int i = 0; i < fft_size; ++i) y[i] = 0.0;
    for (int i = 0; i < x_length; ++i) y[i] = x[i];
    double mean_y = 0.0;
    for (int i = 0; i < y_length; ++i) mean_y += y[i];
    mean_y /= y_length;
    for (int i = 0; i < y_length; ++i) y[i] -= mean_y;
    for (int i = y_length; i < fft_size; ++i) y[i] = 0.0;
    ko_rfft(y, fft_size, y_spectrum);
    int cutoff_in_sample = matlab_round(actual_fs / kCutOff);
    DesignLowCutFilter(cutoff_in_sample * 2 + 1, fft_size, y);
    double *filter_spectrum = dalloc(2 * (fft_size / 2 + 1));
    ko_rfft(y, fft_size, filter_spectrum);
    for (int i = 0; i <= fft_size / 2; ++i) {
      double tmp = y_spectrum[2 * i] * filter_spectrum[2 * i] - y_spectrum[2 * i + 1] * filter_spectrum[2 * i + 1];
      y_spectrum[2 * i + 1] = y_spectrum[2 * i] * filter_spectrum[2 * i + 1] + y_spectrum[2 * i + 1] * filter_spectrum[2 * i];
      y_spectrum[2 * i] = tmp;
    }
    free(filter_spectrum);
  }

  int f0_length = (int)ko_dio_samples(fs, x_length, frame_period);
  for (int i = 0; i < f0_length; ++i) {
    temporal_positions[i] = i * frame_period / 1000.0;
    f0[i] = 0.0;
  }
  double **f0_candidates = (double **)malloc(sizeof(double *) * number_of_bands);
  double **f0_scores = (double **)malloc(sizeof(double *) * number_of_bands);
  double *f0_candidate = dalloc(f0_length), *f0_score = dalloc(f0_length);
  for (int i = 0; i < number_of_bands; ++i) {
    f0_candidates[i] = dalloc(f0_length);
    f0_scores[i] = dalloc(f0_length);
    dio_band(boundary_f0_list[i], actual_fs, y_spectrum, y_length, fft_size, f0_floor,
             f0_ceil, temporal_positions, f0_length, f0_score, f0_candidate);
    for (int j = 0; j < f0_length; ++j) {
      f0_scores[i][j] = f0_score[j] / (f0_candidate[j] + kMySafeGuardMinimum);
      f0_candidates[i][j] = f0_candidate[j];
    }
  }

  double *best_f0_contour = dalloc(f0_length);
  for (int i = 0; i < f0_length; ++i) {
    double tmp = f0_scores[0][i];
    best_f0_contour[i] = f0_candidates[0][i];
    for (int j = 1; j < number_of_bands; ++j) {
      if (tmp > f0_scores[j][i]) {
        tmp = f0_scores[j][i];
        best_f0_contour[i] = f0_candidates[j][i];
      }
    }
  }
  FixF0Contour(frame_period, number_of_bands, f0_candidates, best_f0_contour,
               f0_length, f0_floor, allowed_range, f0);

  for (int i = 0; i < number_of_bands; ++i) { free(f0_candidates[i]); free(f0_scores[i]); }
  free(f0_candidates); free(f0_scores); free(f0_candidate); free(f0_score);
  free(best_f0_contour); free(y); free(y_spectrum); free(boundary_f0_list);
  return 0;
}

/* ---- StoneMask -------------------------------------------------------------- */
static double sm_FixF0(const double *power_spectrum, const double *numerator_i,
                       int fft_size, double fs, double initial_f0, int number_of_harmonics) {
  double numerator = 0.0, denominator = 0.0;
  for (int i = 0; i < number_of_harmonics; ++i) {
    int index = imin(matlab_round(initial_f0 * fft_size / fs * (i + 1)), fft_size / 2);
    double instantaneous_frequency = power_spectrum[index] == 0.0 ? 0.0 :
        (double)index * fs / fft_size + numerator_i[index] / power_spectrum[index] * fs / 2.0 / kPi;
    double amplitude = sqrt(power_spectrum[index]);
    numerator += amplitude * instantaneous_frequency;
    denominator += amplitude * (i + 1);
  }
  return numerator / (denominator + kMySafeGuardMinimum);
}

static double sm_GetRefinedF0(const double *x, int x_length, double fs,
                              double current_position, double initial_f0) {
  if (initial_f0 <= kFloorF0StoneMask || initial_f0 > fs / 12.0) return 0.0;
  int half_window_length = (int)(1.5 * fs / initial_f0 + 1.0);
  double window_length_in_time = (2.0 * half_window_length + 1.0) / fs;
  int base_time_length = half_window_length * 2 + 1;
  int fft_size = (int)pow(2.0, 2.0 + (int)(log(half_window_length * 2.0 + 1.0) / kLog2));
  int half = fft_size / 2;

  double base_time0 = (double)(-half_window_length) / fs;
  int basic_index = matlab_round((current_position + base_time0) * (int)fs + 0.001);
  double *main_window = dalloc(base_time_length), *diff_window = dalloc(base_time_length);
  for (int i = 0; i < base_time_length; ++i) {
    double tmp = (basic_index + i - 1.0) / (int)fs - current_position;
    main_window[i] = 0.42 + 0.5 * cos(2.0 * kPi * tmp / window_length_in_time) +
                     0.08 * cos(4.0 * kPi * tmp / window_length_in_time);
  }
  diff_window[0] = -main_window[1] / 2.0;
  for (int i = 1; i < base_time_length - 1; ++i)
    diff_window[i] = -(main_window[i + 1] - main_window[i - 1]) / 2.0;
  diff_window[base_time_length - 1] = main_window[base_time_length - 2] / 2.0;

  double *wave = dalloc(fft_size);
  double *main_spectrum = dalloc(2 * (half + 1)), *diff_spectrum = dalloc(2 * (half + 1));
  for (int i = 0; i < base_time_length; ++i) {
    int idx = imax(0, imin(x_length - 1, basic_index + i - 1));
    wave[i] = x[idx] * main_window[i];
  }
  for (int i = base_time_length; i < fft_size; ++i) wave[i] = 0.0;
  ko_rfft(wave, fft_size, main_spectrum);
  for (int i = 0; i < base_time_length; ++i) {
    int idx = imax(0, imin(x_length - 1, basic_index + i - 1));
    wave[i] = x[idx] * diff_window[i];
  }
  for (int i = base_time_length; i < fft_size; ++i) wave[i] = 0.0;
  ko_rfft(wave, fft_size, diff_spectrum);

  double *power_spectrum = dalloc(half + 1), *numerator_i = dalloc(half + 1);
  for (int j = 0; j <= half; ++j) {
    numerator_i[j] = main_spectrum[2 * j] * diff_spectrum[2 * j + 1] -
                     main_spectrum[2 * j + 1] * diff_spectrum[2 * j];
    power_spectrum[j] = main_spectrum[2 * j] * main_spectrum[2 * j] +
                        main_spectrum[2 * j + 1] * main_spectrum[2 * j + 1];
  }
  int number_of_harmonics = imin((int)(fs / 2.0 / initial_f0), 6);
  double tentative_f0 = sm_FixF0(power_spectrum, numerator_i, fft_size, fs, initial_f0, 2);
  double mean_f0;
  if (tentative_f0 <= 0.0 || tentative_f0 > initial_f0 * 2)
    mean_f0 = 0.0; /* the fixed value is too large: rejected */
  else
    mean_f0 = sm_FixF0(power_spectrum, numerator_i, fft_size, fs, tentative_f0, number_of_harmonics);

  free(main_window); free(diff_window); free(wave); free(main_spectrum);
  free(diff_spectrum); free(power_spectrum); free(numerator_i);
  /* if the amount of correction is overlarge (20 %), the initial F0 is kept */
  if (fabs(mean_f0 - initial_f0) > initial_f0 * 0.2) mean_f0 = initial_f0;
  return mean_f0;
}

int ko_stonemask(const double *x, int64_t x_length, int fs, const double *t,
                 const double *f0, int64_t f0_length, double *refined_f0) {
  for (int64_t i = 0; i < f0_length; ++i)
    refined_f0[i] = sm_GetRefinedF0(x, (int)x_length, fs, t[i], f0[i]);
  return 0;
}
