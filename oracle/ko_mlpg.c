/*
 * oracle/ko_mlpg.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C restatement of the converter-apply arithmetic the reference delegates to
 * nnmnkwii 0.0.17 (+ bandmat 0.7, scikit-learn) (Pipfile.lock:86,19,165):
 *
 *   delta_features(x, windows)            kwiiyatta/converter/delta.py:30,46
 *   MLPG(gmm, windows, diff).transform(X) kwiiyatta/converter/gmm.py:28-34
 *
 * nnmnkwii.baseline.gmm.MLPG.transform, restated:
 *   1. m_t = argmax_m  log w_m + log N(x_t | mu_x[m], S_xx[m])   (px.predict)
 *   2. E_t = mu_y[m_t] + S_yx[m_t] S_xx[m_t]^-1 (x_t - mu_x[m_t])
 *   3. D_t = diag(S_yy[m_t]) - diag(S_yx[m_t]) / diag(S_xx[m_t]) * diag(S_xy[m_t])
 *      (nnmnkwii approximates the covariances as diagonals here)
 *   4. nnmnkwii.paramgen.mlpg(E, D, windows): per static dimension, solve
 *      (sum_w W_w' diag(1/D_w) W_w) y = sum_w W_w' (E_w / D_w)   by banded Cholesky.
 *   diff=True first rewrites mu_y -= mu_x; S_yy = S_xx+S_yy-S_xy-S_yx;
 *   S_xy -= S_xx; S_yx = S_xy'.
 *
 * None of those packages is vendored under /root/reference or installed here
 * (scikit-learn is, and generates the GMM fixtures): published algorithm
 * restated, parity with upstream unpinned except through the end-to-end KAT
 * tests/kwiiyatta/test_convert_voice.py:118-129.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

/* y[t] = sum_k x[t+k] w[k+l], zero outside (np.correlate(x, w, 'same')) */
int ko_delta_features(const double *x, int64_t T, int d, int nwin, const int *wl,
                      const int *wu, const double *wcoef /* nwin x KO_MAX_WIN */,
                      double *out /* T x (nwin*d) */) {
  int D = nwin * d;
  for (int w = 0; w < nwin; ++w)
    for (int64_t t = 0; t < T; ++t)
      for (int c = 0; c < d; ++c) {
        double s = 0.0;
        for (int k = -wl[w]; k <= wu[w]; ++k) {
          int64_t tt = t + k;
          if (tt < 0 || tt >= T) continue;
          s += x[tt * d + c] * wcoef[w * KO_MAX_WIN + k + wl[w]];
        }
        out[t * D + w * d + c] = s;
      }
  return 0;
}

/* dense Cholesky A = L L' (lower), in place on a copy; returns 0 on success */
static int chol(double *a, int n) {
  for (int j = 0; j < n; ++j) {
    double s = a[j * n + j];
    for (int k = 0; k < j; ++k) s -= a[j * n + k] * a[j * n + k];
    if (s <= 0.0) return -1;
    double l = sqrt(s);
    a[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = a[i * n + j];
      for (int k = 0; k < j; ++k) v -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = v / l;
    }
  }
  return 0;
}

/* solve the banded SPD system (half bandwidth bw) P y = b; P stored as
 * rows of (bw+1) lower-band entries: Pb[t*(bw+1)+k] = P[t][t-k]. */
static void banded_solveh(double *Pb, double *b, int64_t T, int bw) {
  int s = bw + 1;
  /* banded Cholesky: L[t][t-k] overwrites Pb */
  for (int64_t t = 0; t < T; ++t) {
    for (int k = bw; k >= 0; --k) {
      int64_t c = t - k;
      if (c < 0) continue;
      double v = Pb[t * s + k];
      /* subtract sum_{m<c} L[t][m] L[c][m], m >= t-bw */
      for (int64_t m = (t - bw > 0 ? t - bw : 0); m < c; ++m)
        if (c - m <= bw) v -= Pb[t * s + (t - m)] * Pb[c * s + (c - m)];
      if (k == 0) Pb[t * s] = sqrt(v);
      else Pb[t * s + k] = v / Pb[c * s];
    }
  }
  /* forward L z = b */
  for (int64_t t = 0; t < T; ++t) {
    double v = b[t];
    for (int k = 1; k <= bw && t - k >= 0; ++k) v -= Pb[t * s + k] * b[t - k];
    b[t] = v / Pb[t * s];
  }
  /* backward L' y = z */
  for (int64_t t = T - 1; t >= 0; --t) {
    double v = b[t];
    for (int k = 1; k <= bw && t + k < T; ++k) v -= Pb[(t + k) * s + k] * b[t + k];
    b[t] = v / Pb[t * s];
  }
}

int ko_mlpg(const double *mean_frames, const double *variance_frames, int64_t T,
            int static_dim, int nwin, const int *wl, const int *wu,
            const double *wcoef, double *y /* T x static_dim */) {
  int D = nwin * static_dim;
  int bw = 0;
  for (int w = 0; w < nwin; ++w) if (wl[w] + wu[w] > bw) bw = wl[w] + wu[w];
  int s = bw + 1;
  double *Pb = (double *)malloc(sizeof(double) * T * s);
  double *b = (double *)malloc(sizeof(double) * T);
  for (int d = 0; d < static_dim; ++d) {
    memset(Pb, 0, sizeof(double) * T * s);
    memset(b, 0, sizeof(double) * T);
    for (int w = 0; w < nwin; ++w) {
      for (int64_t t = 0; t < T; ++t) {
        double prec = 1 / variance_frames[t * D + w * static_dim + d];
        double bm = prec * mean_frames[t * D + w * static_dim + d];
        for (int k1 = -wl[w]; k1 <= wu[w]; ++k1) {
          int64_t a = t + k1;
          if (a < 0 || a >= T) continue;
          double ca = wcoef[w * KO_MAX_WIN + k1 + wl[w]];
          b[a] += ca * bm;
          for (int k2 = -wl[w]; k2 <= k1; ++k2) {
            int64_t c = t + k2;
            if (c < 0 || c >= T) continue;
            double cc = wcoef[w * KO_MAX_WIN + k2 + wl[w]];
            Pb[a * s + (a - c)] += ca * prec * cc;
          }
        }
      }
    }
    banded_solveh(Pb, b, T, bw);
    for (int64_t t = 0; t < T; ++t) y[t * static_dim + d] = b[t];
  }
  free(Pb); free(b);
  return 0;
}

int ko_gmm_mlpg(const double *x /* T x d static */, int64_t T, int d, int M,
                const double *weights, const double *means /* M x 2D */,
                const double *covs /* M x 2D x 2D */, int diff, int nwin,
                const int *wl, const int *wu, const double *wcoef,
                double *y /* T x d */, int32_t *mix_out /* T or NULL */) {
  int D = nwin * d, D2 = 2 * D;
  double *X = (double *)malloc(sizeof(double) * T * D);
  ko_delta_features(x, T, d, nwin, wl, wu, wcoef, X);

  /* split the joint GMM (MLPGBase.__init__) */
  double *mux = (double *)malloc(sizeof(double) * M * D);
  double *muy = (double *)malloc(sizeof(double) * M * D);
  double *Sxx = (double *)malloc(sizeof(double) * M * D * D);
  double *Sxy = (double *)malloc(sizeof(double) * M * D * D);
  double *Syx = (double *)malloc(sizeof(double) * M * D * D);
  double *Syy = (double *)malloc(sizeof(double) * M * D * D);
  for (int m = 0; m < M; ++m) {
    for (int i = 0; i < D; ++i) {
      mux[m * D + i] = means[m * D2 + i];
      muy[m * D + i] = means[m * D2 + D + i];
      for (int j = 0; j < D; ++j) {
        const double *C = covs + (size_t)m * D2 * D2;
        Sxx[(m * D + i) * D + j] = C[i * D2 + j];
        Sxy[(m * D + i) * D + j] = C[i * D2 + D + j];
        Syx[(m * D + i) * D + j] = C[(D + i) * D2 + j];
        Syy[(m * D + i) * D + j] = C[(D + i) * D2 + D + j];
      }
    }
  }
  if (diff) {
    for (int m = 0; m < M; ++m) {
      for (int i = 0; i < D; ++i) muy[m * D + i] -= mux[m * D + i];
      for (int i = 0; i < D * D; ++i) {
        size_t k = (size_t)m * D * D + i;
        Syy[k] = Sxx[k] + Syy[k] - Sxy[k] - Syx[k];
      }
      for (int i = 0; i < D * D; ++i) {
        size_t k = (size_t)m * D * D + i;
        Sxy[k] = Sxy[k] - Sxx[k];
      }
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
          Syx[(m * D + i) * D + j] = Sxy[(m * D + j) * D + i];
    }
  }

  /* Cholesky of S_xx per mixture and log-det term */
  double *L = (double *)malloc(sizeof(double) * M * D * D);
  double *logdet = (double *)malloc(sizeof(double) * M);
  for (int m = 0; m < M; ++m) {
    memcpy(L + (size_t)m * D * D, Sxx + (size_t)m * D * D, sizeof(double) * D * D);
    if (chol(L + (size_t)m * D * D, D) != 0) return -2;
    double ld = 0.0;
    for (int i = 0; i < D; ++i) ld -= log(L[(size_t)m * D * D + i * D + i]);
    logdet[m] = ld; /* = sum log diag(precisions_cholesky) */
  }

  double *E = (double *)malloc(sizeof(double) * T * D);
  double *Dv = (double *)malloc(sizeof(double) * T * D);
  double *z = (double *)malloc(sizeof(double) * D);
  double *zbest = (double *)malloc(sizeof(double) * D);
  for (int64_t t = 0; t < T; ++t) {
    const double *xt = X + t * D;
    int best_m = 0; double best = -INFINITY;
    for (int m = 0; m < M; ++m) {
      const double *Lm = L + (size_t)m * D * D;
      double q = 0.0;
      for (int i = 0; i < D; ++i) {
        double v = xt[i] - mux[m * D + i];
        for (int k = 0; k < i; ++k) v -= Lm[i * D + k] * z[k];
        z[i] = v / Lm[i * D + i];
        q += z[i] * z[i];
      }
      double lp = -0.5 * (D * log(2.0 * M_PI) + q) + logdet[m] + log(weights[m]);
      if (lp > best) { best = lp; best_m = m; memcpy(zbest, z, sizeof(double) * D); }
    }
    int m = best_m;
    if (mix_out) mix_out[t] = m;
    /* xx = S_xx^-1 (x - mu_x): back substitution L' xx = z */
    const double *Lm = L + (size_t)m * D * D;
    for (int i = D - 1; i >= 0; --i) {
      double v = zbest[i];
      for (int k = i + 1; k < D; ++k) v -= Lm[k * D + i] * z[k];
      z[i] = v / Lm[i * D + i];
    }
    for (int i = 0; i < D; ++i) {
      double v = muy[m * D + i];
      for (int k = 0; k < D; ++k) v += Syx[(m * D + i) * D + k] * z[k];
      E[t * D + i] = v;
      Dv[t * D + i] = Syy[(m * D + i) * D + i] -
                      Syx[(m * D + i) * D + i] / Sxx[(m * D + i) * D + i] * Sxy[(m * D + i) * D + i];
    }
  }
  ko_mlpg(E, Dv, T, d, nwin, wl, wu, wcoef, y);

  free(X); free(mux); free(muy); free(Sxx); free(Sxy); free(Syx); free(Syy);
  free(L); free(logdet); free(E); free(Dv); free(z); free(zbest);
  return 0;
}
