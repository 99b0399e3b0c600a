/*
 * oracle/ko_dtw.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C restatement of fastdtw 0.3.2 (Pipfile.lock:68), pure-Python flavour
 * (fastdtw/fastdtw.py), as called at kwiiyatta/vocoder/align.py:71:
 *
 *   fastdtw(x, y, radius, dist=2)
 *     - len(x) < radius+2 or len(y) < radius+2  -> full DTW
 *     - else halve both series ((x[2i]+x[2i+1])/2, odd tail dropped),
 *       recurse, expand the coarse path by +-radius, double the resolution,
 *       and run the DP restricted to that window.
 *   DP: D[i,j] = d(x_i,y_j) + min(D[i-1,j], D[i,j-1], D[i-1,j-1]); ties are
 *   resolved in THAT order (Python min(..., key=) keeps the first minimum),
 *   comparing the sums D+d as the Python code does.
 *
 * Window note: the coarse path is monotone, so for every row the expanded
 * window is one contiguous column range; the scan in __expand_window then
 * returns exactly that range.  This file computes the range directly.
 *
 * fastdtw is not vendored under /root/reference nor installed here; parity
 * with upstream is pinned only by the KAT tests/kwiiyatta/test_vocoder.py:266-288.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

typedef struct { int32_t i, j; } cell;

static double dist2(const double *a, const double *b, int dim) {
  double s = 0.0;
  for (int k = 0; k < dim; ++k) { double d = a[k] - b[k]; s += d * d; }
  return sqrt(s);
}

/* DP over per-row windows [lo[i], hi[i]] (inclusive, 0-based); returns path
 * length; path written front to back. */
static int64_t windowed_dtw(const double *x, int len_x, const double *y, int len_y,
                            int dim, const int *lo, const int *hi, double *out_dist,
                            cell *path) {
  /* cost / predecessor storage per row, 1-based DP indices as in fastdtw */
  int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (len_x + 1));
  int64_t total = 0;
  for (int i = 0; i < len_x; ++i) { off[i] = total; total += hi[i] - lo[i] + 1; }
  off[len_x] = total;
  double *D = (double *)malloc(sizeof(double) * (total ? total : 1));
  unsigned char *P = (unsigned char *)malloc(total ? total : 1);

#define DGET(i, j) (((i) >= 0 && (j) >= lo[(i)] && (j) <= hi[(i)]) ? D[off[(i)] + (j) - lo[(i)]] : INFINITY)
  for (int i = 0; i < len_x; ++i) {
    for (int j = lo[i]; j <= hi[i]; ++j) {
      double dt = dist2(x + (int64_t)i * dim, y + (int64_t)j * dim, dim);
      double up, left, diag; /* D[i-1,j], D[i,j-1], D[i-1,j-1] in 0-based cells */
      if (i == 0 && j == 0) { up = INFINITY; left = INFINITY; diag = 0.0; }
      else {
        up = i > 0 ? DGET(i - 1, j) : INFINITY;
        left = j > 0 ? DGET(i, j - 1) : INFINITY;
        diag = (i > 0 && j > 0) ? DGET(i - 1, j - 1) : INFINITY;
      }
      double c0 = up + dt, c1 = left + dt, c2 = diag + dt;
      double best = c0; unsigned char pb = 0;
      if (c1 < best) { best = c1; pb = 1; }
      if (c2 < best) { best = c2; pb = 2; }
      D[off[i] + j - lo[i]] = best;
      P[off[i] + j - lo[i]] = pb;
    }
  }
  *out_dist = DGET(len_x - 1, len_y - 1);
  /* back-trace */
  int64_t n = 0;
  int i = len_x - 1, j = len_y - 1;
  while (i >= 0 && j >= 0) {
    path[n].i = i; path[n].j = j; ++n;
    if (i == 0 && j == 0) break;
    unsigned char pb = (j >= lo[i] && j <= hi[i]) ? P[off[i] + j - lo[i]] : 0;
    if (pb == 0) --i; else if (pb == 1) --j; else { --i; --j; }
  }
#undef DGET
  for (int64_t a = 0, b = n - 1; a < b; ++a, --b) { cell t = path[a]; path[a] = path[b]; path[b] = t; }
  free(off); free(D); free(P);
  return n;
}

static int64_t fastdtw_rec(const double *x, int len_x, const double *y, int len_y,
                           int dim, int radius, double *out_dist, cell *path) {
  int min_time_size = radius + 2;
  int *lo = (int *)malloc(sizeof(int) * len_x);
  int *hi = (int *)malloc(sizeof(int) * len_x);
  if (len_x < min_time_size || len_y < min_time_size) {
    for (int i = 0; i < len_x; ++i) { lo[i] = 0; hi[i] = len_y - 1; }
  } else {
    int hx = len_x / 2, hy = len_y / 2;
    double *xs = (double *)malloc(sizeof(double) * (size_t)hx * dim);
    double *ys = (double *)malloc(sizeof(double) * (size_t)hy * dim);
    for (int i = 0; i < hx; ++i)
      for (int k = 0; k < dim; ++k)
        xs[(int64_t)i * dim + k] = (x[(int64_t)(2 * i) * dim + k] + x[(int64_t)(2 * i + 1) * dim + k]) / 2;
    for (int i = 0; i < hy; ++i)
      for (int k = 0; k < dim; ++k)
        ys[(int64_t)i * dim + k] = (y[(int64_t)(2 * i) * dim + k] + y[(int64_t)(2 * i + 1) * dim + k]) / 2;
    cell *cpath = (cell *)malloc(sizeof(cell) * (size_t)(hx + hy + 1));
    double cd;
    int64_t cn = fastdtw_rec(xs, hx, ys, hy, dim, radius, &cd, cpath);
    free(xs); free(ys);
    /* coarse rows a = 0 .. (len_x-1)/2: columns [min pj - r, max pj + r] over
     * path cells with |pi - a| <= r */
    int ca = (len_x - 1) / 2 + 1;
    int *clo = (int *)malloc(sizeof(int) * ca), *chi = (int *)malloc(sizeof(int) * ca);
    for (int a = 0; a < ca; ++a) { clo[a] = 1 << 30; chi[a] = -(1 << 30); }
    for (int64_t k = 0; k < cn; ++k) {
      int a0 = cpath[k].i - radius, a1 = cpath[k].i + radius;
      if (a0 < 0) a0 = 0;
      if (a1 > ca - 1) a1 = ca - 1;
      for (int a = a0; a <= a1; ++a) {
        if (cpath[k].j - radius < clo[a]) clo[a] = cpath[k].j - radius;
        if (cpath[k].j + radius > chi[a]) chi[a] = cpath[k].j + radius;
      }
    }
    for (int i = 0; i < len_x; ++i) {
      int a = i / 2;
      int l = 2 * clo[a], h = 2 * chi[a] + 1;
      if (l < 0) l = 0;
      if (h > len_y - 1) h = len_y - 1;
      lo[i] = l; hi[i] = h;
    }
    free(clo); free(chi); free(cpath);
  }
  int64_t n = windowed_dtw(x, len_x, y, len_y, dim, lo, hi, out_dist, path);
  free(lo); free(hi);
  return n;
}

int ko_fastdtw(const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
               int radius, double *dist, int32_t *path, int64_t *path_len) {
  cell *p = (cell *)malloc(sizeof(cell) * (size_t)(Tx + Ty + 1));
  int64_t n = fastdtw_rec(x, (int)Tx, y, (int)Ty, dim, radius, dist, p);
  for (int64_t k = 0; k < n; ++k) { path[2 * k] = p[k].i; path[2 * k + 1] = p[k].j; }
  *path_len = n;
  free(p);
  return 0;
}
