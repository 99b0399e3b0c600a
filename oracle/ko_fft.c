/*
 * oracle/ko_fft.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See ko_fft.h.
 *
 * Iterative radix-2 decimation-in-time complex FFT with cached twiddle and
 * bit-reversal tables; real transforms go through a half-length complex FFT.
 * Single-threaded; tables are created lazily (not thread-safe by design: the
 * oracle is driven from one thread per process).
 */
#include "ko_fft.h"

#include <math.h>
#include <stdlib.h>

#define KO_MAX_LOG2 24

typedef struct {
  int n;
  double *tw;   /* n/2 complex twiddles exp(-2 pi i k / n) */
  int *rev;     /* bit reversal permutation */
} ko_plan;

static ko_plan g_plans[KO_MAX_LOG2 + 1];

static int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

static const ko_plan *get_plan(int n) {
  int l = ilog2(n);
  ko_plan *p = &g_plans[l];
  if (p->n == n) return p;
  p->n = n;
  p->tw = (double *)malloc(sizeof(double) * (n > 1 ? n : 2));
  p->rev = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n / 2; ++k) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    p->tw[2 * k] = cos(a);
    p->tw[2 * k + 1] = sin(a);
  }
  p->rev[0] = 0;
  for (int i = 1; i < n; ++i)
    p->rev[i] = (p->rev[i >> 1] >> 1) | ((i & 1) ? (n >> 1) : 0);
  return p;
}

void ko_cfft(double *z, int n, int sign) {
  if (n <= 1) return;
  const ko_plan *p = get_plan(n);
  for (int i = 0; i < n; ++i) {
    int j = p->rev[i];
    if (j > i) {
      double tr = z[2 * i], ti = z[2 * i + 1];
      z[2 * i] = z[2 * j];
      z[2 * i + 1] = z[2 * j + 1];
      z[2 * j] = tr;
      z[2 * j + 1] = ti;
    }
  }
  const double s = sign < 0 ? 1.0 : -1.0; /* conj twiddle for backward */
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1;
    int step = n / len;
    for (int i = 0; i < n; i += len) {
      for (int k = 0; k < half; ++k) {
        double wr = p->tw[2 * k * step];
        double wi = s * p->tw[2 * k * step + 1];
        double *a = z + 2 * (i + k);
        double *b = z + 2 * (i + k + half);
        double xr = b[0] * wr - b[1] * wi;
        double xi = b[0] * wi + b[1] * wr;
        b[0] = a[0] - xr;
        b[1] = a[1] - xi;
        a[0] += xr;
        a[1] += xi;
      }
    }
  }
}

void ko_rfft(const double *x, int n, double *out) {
  if (n == 1) { out[0] = x[0]; out[1] = 0.0; return; }
  if (n == 2) {
    out[0] = x[0] + x[1]; out[1] = 0.0;
    out[2] = x[0] - x[1]; out[3] = 0.0;
    return;
  }
  int h = n / 2;
  double *z = (double *)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) z[i] = x[i]; /* z[k] = x[2k] + i x[2k+1] */
  ko_cfft(z, h, -1);
  const ko_plan *p = get_plan(n);
  /* X[k] = E[k] + w^k O[k],  E = (Z[k] + conj Z[h-k])/2, O = (Z[k] - conj Z[h-k])/(2i) */
  out[0] = z[0] + z[1];
  out[1] = 0.0;
  out[2 * h] = z[0] - z[1];
  out[2 * h + 1] = 0.0;
  for (int k = 1; k < h; ++k) {
    double ar = z[2 * k], ai = z[2 * k + 1];
    double br = z[2 * (h - k)], bi = -z[2 * (h - k) + 1];
    double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
    /* O = d / i = (di, -dr) */
    double or_ = di, oi = -dr;
    double wr = p->tw[2 * k], wi = p->tw[2 * k + 1];
    out[2 * k] = er + (or_ * wr - oi * wi);
    out[2 * k + 1] = ei + (or_ * wi + oi * wr);
  }
  free(z);
}

void ko_irfft(const double *spec, int n, double *x) {
  if (n == 1) { x[0] = spec[0]; return; }
  if (n == 2) {
    x[0] = spec[0] + spec[2];
    x[1] = spec[0] - spec[2];
    return;
  }
  int h = n / 2;
  const ko_plan *p = get_plan(n);
  double *z = (double *)malloc(sizeof(double) * n);
  /* Z[k] = E[k] + i O[k], with E[k] = X[k] + conj X[h-k], O[k] = (X[k] - conj X[h-k]) conj(w^k)
   * (factor 2 absorbed: result is n * true inverse, i.e. unnormalised c2r). */
  for (int k = 0; k < h; ++k) {
    double ar = spec[2 * k], ai = (k == 0) ? 0.0 : spec[2 * k + 1];
    double br = spec[2 * (h - k)], bi = (k == 0) ? 0.0 : -spec[2 * (h - k) + 1];
    double er = ar + br, ei = ai + bi;
    double dr = ar - br, di = ai - bi;
    double wr = p->tw[2 * k], wi = -p->tw[2 * k + 1];
    double or_ = dr * wr - di * wi, oi = dr * wi + di * wr;
    z[2 * k] = er - oi;
    z[2 * k + 1] = ei + or_;
  }
  ko_cfft(z, h, +1);
  for (int i = 0; i < n; ++i) x[i] = z[i];
  free(z);
}
