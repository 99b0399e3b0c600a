/*
 * oracle/ko_sptk.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C restatement of the pysptk 0.1.16 (Pipfile.lock:147) conversions the
 * reference calls at kwiiyatta/vocoder/mcep.py:65 (mc2sp) and :71 (sp2mc):
 *
 *   sp2mc(P, order, a): c = irfft(log P); c[0] /= 2; mc = freqt(c, order, a)
 *   mc2sp(mc, a, fftlen): c = freqt(mc, fftlen/2, -a); c[0] *= 2;
 *                         mirror to fftlen; exp(real(rfft(c)))
 *
 * with SPTK's `freqt` frequency-transformation recursion.  pysptk is not
 * vendored under /root/reference nor installed here: published algorithm
 * restated; parity with upstream pinned only by the reference's envelope
 * KAT tests/kwiiyatta/test_vocoder.py:233-244.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_fft.h"
#include "ko_oracle.h"

/* SPTK freqt: c1[0..m1] -> c2[0..m2], all-pass constant a. */
void ko_freqt(const double *c1, int m1, double *c2, int m2, double a) {
  double *d = (double *)calloc(m2 + 1, sizeof(double));
  double *g = (double *)calloc(m2 + 1, sizeof(double));
  double b = 1 - a * a;
  for (int i = -m1; i <= 0; ++i) {
    if (0 <= m2) { d[0] = g[0]; g[0] = c1[-i] + a * d[0]; }
    if (1 <= m2) { d[1] = g[1]; g[1] = b * d[0] + a * d[1]; }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
  memcpy(c2, g, sizeof(double) * (m2 + 1));
  free(d); free(g);
}

int ko_sp2mc(const double *sp, int64_t T, int K, int order, double alpha, double *mc) {
  int n = 2 * (K - 1);
  double *spec = (double *)malloc(sizeof(double) * 2 * K);
  double *c = (double *)malloc(sizeof(double) * n);
  for (int64_t t = 0; t < T; ++t) {
    const double *p = sp + t * K;
    for (int k = 0; k < K; ++k) { spec[2 * k] = log(p[k]); spec[2 * k + 1] = 0.0; }
    ko_irfft(spec, n, c);
    for (int i = 0; i < n; ++i) c[i] /= n; /* numpy irfft normalisation */
    c[0] /= 2.0;
    ko_freqt(c, n - 1, mc + t * (order + 1), order, alpha);
  }
  free(spec); free(c);
  return 0;
}

int ko_mc2sp(const double *mc, int64_t T, int order, double alpha, int fftlen, double *sp) {
  int half = fftlen / 2, K = half + 1;
  double *c = (double *)malloc(sizeof(double) * (half + 1));
  double *symc = (double *)malloc(sizeof(double) * fftlen);
  double *spec = (double *)malloc(sizeof(double) * 2 * K);
  for (int64_t t = 0; t < T; ++t) {
    ko_freqt(mc + t * (order + 1), order, c, half, -alpha);
    c[0] *= 2.0;
    memset(symc, 0, sizeof(double) * fftlen);
    symc[0] = c[0];
    for (int i = 1; i <= half; ++i) {
      symc[i] = c[i];
      symc[fftlen - i] = c[i];
    }
    ko_rfft(symc, fftlen, spec);
    for (int k = 0; k < K; ++k) sp[t * K + k] = exp(spec[2 * k]);
  }
  free(c); free(symc); free(spec);
  return 0;
}
