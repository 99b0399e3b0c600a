/*
 * oracle/ko_mlsa.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C restatement of the pysptk 0.1.16 (Pipfile.lock:147) routines behind the reference's
 * differential-spectrum filter, kwiiyatta/filter/mlsa.py:9-30:
 *
 *     b = pysptk.mc2b(mc, alpha)
 *     Synthesizer(MLSADF(order, alpha), hopsize).synthesis(wav, b)
 *
 *   mc2b       b[m] = c[m];  b[i] = c[i] - a b[i+1]                         (SPTK mc2b)
 *   MLSADF     SPTK's mlsadf(): the exponential transfer function exp F(z) as a Pade
 *              approximant of order pd (pysptk default pd = 4), two cascaded sections:
 *              mlsadf1 carries b[1] on a first-order all-pass, mlsadf2 carries b[2..m] on the
 *              all-pass chain of mlsafir(); delay line of 3(pd+1) + pd(m+2) doubles
 *   synthesis  pysptk.synthesis.Synthesizer: frame i covers samples [i hop, (i+1) hop); inside a
 *              frame the coefficients move linearly from the previous frame's to this frame's
 *              (slope = (cur - prev)/hop, added after every sample; the first frame starts from
 *              its own coefficients); the input sample is scaled by exp(b[0]); a frame whose
 *              end reaches the end of the signal is not processed.  Upstream allocates the
 *              output with np.empty_like, i.e. the unprocessed tail is undefined there; here it
 *              is zero.
 *
 * pysptk is not vendored under /root/reference nor installed here: the published algorithm
 * (Imai's MLSA filter as implemented in SPTK 3.x) restated from the SPTK sources' structure.
 * Pinned by a property of the algorithm, not by upstream vectors: white noise through the filter
 * of a constant mel-cepstrum acquires the spectrum mc2sp(mc) (tests/test_mlsa.py), and by the
 * reference's envelope KAT tests/kwiiyatta/test_filter.py:45-48 only through the whole stack.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

static const double kPade[] = {1.0,
                               1.0, 0.0,
                               1.0, 0.0, 0.0,
                               1.0, 0.0, 0.0, 0.0,
                               1.0, 0.4999273, 0.1067005, 0.01170221, 0.0005656279,
                               1.0, 0.4999391, 0.1107098, 0.01369984, 0.0009564853, 0.00003041721};

void ko_mc2b(const double *mc, int64_t T, int m, double a, double *b) {
  for (int64_t t = 0; t < T; ++t) {
    const double *c = mc + t * (m + 1);
    double *o = b + t * (m + 1);
    o[m] = c[m];
    for (int i = m - 1; i >= 0; --i) o[i] = c[i] - a * o[i + 1];
  }
}

static double mlsafir(double x, const double *b, int m, double a, double *d) {
  double y = 0.0;
  const double aa = 1 - a * a;
  d[0] = x;
  d[1] = aa * d[0] + a * d[1];
  for (int i = 2; i <= m; ++i) {
    d[i] = d[i] + a * (d[i + 1] - d[i - 1]);
    y += d[i] * b[i];
  }
  for (int i = m + 1; i > 1; --i) d[i] = d[i - 1];
  return y;
}

static double mlsadf1(double x, const double *b, double a, int pd, double *d, const double *ppade) {
  double v, out = 0.0;
  const double aa = 1 - a * a;
  double *pt = &d[pd + 1];
  for (int i = pd; i >= 1; --i) {
    d[i] = aa * pt[i - 1] + a * d[i];
    pt[i] = d[i] * b[1];
    v = pt[i] * ppade[i];
    x += (1 & i) ? v : -v;
    out += v;
  }
  pt[0] = x;
  out += x;
  return out;
}

static double mlsadf2(double x, const double *b, int m, double a, int pd, double *d, const double *ppade) {
  double v, out = 0.0;
  double *pt = &d[pd * (m + 2)];
  for (int i = pd; i >= 1; --i) {
    pt[i] = mlsafir(pt[i - 1], b, m, a, &d[(i - 1) * (m + 2)]);
    v = pt[i] * ppade[i];
    x += (1 & i) ? v : -v;
    out += v;
  }
  pt[0] = x;
  out += x;
  return out;
}

int ko_mlsadf_delay_length(int m, int pd) { return 3 * (pd + 1) + pd * (m + 2); }

double ko_mlsadf(double x, const double *b, int m, double a, int pd, double *d) {
  const double *ppade = &kPade[pd * (pd + 1) / 2];
  x = mlsadf1(x, b, a, pd, d, ppade);
  x = mlsadf2(x, b, m, a, pd, &d[2 * (pd + 1)], ppade);
  return x;
}

/* Synthesizer(MLSADF(m, a, pd), hop).synthesis(x, b): b is T x (m+1) */
int ko_mlsa_synthesis(const double *x, int64_t n, const double *b, int64_t T, int m, double a, int pd,
                      int hop, double *y) {
  if (pd < 4 || pd > 5 || m < 1 || hop < 1) return -1;
  double *d = (double *)calloc(ko_mlsadf_delay_length(m, pd), sizeof(double));
  double *cur = (double *)malloc(sizeof(double) * (m + 1));
  double *slope = (double *)malloc(sizeof(double) * (m + 1));
  memset(y, 0, sizeof(double) * n);
  const double *prev = b;
  for (int64_t i = 0; i < T; ++i) {
    const int64_t s = i * hop, e = (i + 1) * hop;
    if (e >= n) break;
    const double *curb = b + i * (m + 1);
    for (int k = 0; k <= m; ++k) { slope[k] = (curb[k] - prev[k]) / hop; cur[k] = prev[k]; }
    for (int64_t j = s; j < e; ++j) {
      const double scaled = x[j] * exp(cur[0]);
      y[j] = ko_mlsadf(scaled, cur, m, a, pd, d);
      for (int k = 0; k <= m; ++k) cur[k] += slope[k];
    }
    prev = curb;
  }
  free(d); free(cur); free(slope);
  return 0;
}
