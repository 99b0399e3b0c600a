// kwy_gmmfit.hip -- EM for a full-covariance Gaussian mixture on gfx950: the building
// blocks of sklearn.mixture.GaussianMixture.fit as the reference uses it
// (kwiiyatta/converter/gmm.py:14-26: n_components=64, covariance_type='full',
// reg_covar=1e-6, tol=1e-3, max_iter=100 on the joint static+delta features, D = 144).
//
// One EM iteration, data-parallel over frames (each GPU holds a shard of X):
//   E:  k_fit_prec      per mixture: packed Cholesky of the covariance in LDS, its inverse,
//                       log-normaliser                                   (replicated)
//       k_fit_logprob   (frame tile x mixture): ||L^-1 (x - mu)||^2 from LDS-resident L^-1
//       k_fit_resp      per frame: logsumexp over mixtures -> responsibilities, log-likelihood
//   M:  k_fit_sums      per row chunk: sum_t r, sum_t r x                  (local statistics)
//       k_fit_cov       (mixture x row split): sum_t r (x-mu)(x-mu)'       (local statistics)
//   The driver (kwiiyatta_amd/converter/gmm_fit.py) all-reduces the statistics over RCCL
//   between the two M-step kernels and after them, then calls kwy_gmm_em_finalize_dev.
//
// FP64 throughout (sklearn semantics: centred two-pass covariance, nk + 10 eps, reg_covar on
// the diagonal).  The two heavy kernels are register/LDS-tiled f64 FMA loops (the f64 MFMA
// rate of gfx950 equals its f64 vector rate).
#include <math.h>

#include "kwy_internal.hpp"

#define FIT_NT 512            // threads of the log-prob workgroup
#define FIT_TILE 64           // frames per log-prob workgroup
#define FIT_SUM_ROWS 64       // rows per LDS tile in k_fit_sums
#define FIT_COV_ROWS 32       // rows per LDS tile in k_fit_cov
#define FIT_COV_SPLIT 8       // row splits per mixture in k_fit_cov

__host__ __device__ static inline size_t fit_tri(int D) { return (size_t)D * (D + 1) / 2; }
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // j <= i

// per mixture: Zp[m] = packed lower-triangular inverse of chol(cov_m), cst[m] = log w - 0.5 D log 2pi - sum log L_ii
__global__ __launch_bounds__(KWY_THREADS) void k_fit_prec(const double *__restrict__ weights,
                                                         const double *__restrict__ covs, int D,
                                                         double *__restrict__ Zp, double *__restrict__ zcol,
                                                         double *__restrict__ cst, int *__restrict__ status) {
  extern __shared__ double L[];  // packed lower triangle
  const int tid = threadIdx.x, m = blockIdx.x;
  const double *C = covs + (size_t)m * D * D;
  const int nt = (int)fit_tri(D);
  for (int i = tid; i < D; i += KWY_THREADS)
    for (int j = 0; j <= i; ++j) L[tri(i, j)] = C[(size_t)i * D + j];
  __syncthreads();
  for (int j = 0; j < D; ++j) {
    const double piv = L[tri(j, j)];
    if (!(piv > 0.0)) { if (tid == 0) atomicExch(status, 1); return; }
    const double dj = sqrt(piv);
    __syncthreads();
    if (tid == 0) L[tri(j, j)] = dj;
    for (int i = j + 1 + tid; i < D; i += KWY_THREADS) L[tri(i, j)] = L[tri(i, j)] / dj;
    __syncthreads();
    const int rem = D - j - 1;
    for (int e = tid; e < rem * rem; e += KWY_THREADS) {
      int i = j + 1 + e / rem, k = j + 1 + e % rem;
      if (k <= i) L[tri(i, k)] -= L[tri(i, j)] * L[tri(k, j)];
    }
    __syncthreads();
  }
  // Z = L^-1, column c by thread c (its column lives in global scratch zcol[m][c][.])
  double *zc_all = zcol + (size_t)m * D * D;
  for (int c = tid; c < D; c += KWY_THREADS) {
    double *zc = zc_all + (size_t)c * D;
    for (int i = c; i < D; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) v -= L[tri(i, k)] * zc[k];
      zc[i] = v / L[tri(i, i)];
    }
  }
  __syncthreads();
  double *zp = Zp + (size_t)m * nt;
  for (int e = tid; e < D * D; e += KWY_THREADS) {
    int i = e / D, c = e % D;
    if (c <= i) zp[tri(i, c)] = zc_all[(size_t)c * D + i];
  }
  if (tid == 0) {
    double ld = 0.0;
    for (int i = 0; i < D; ++i) ld -= log(L[tri(i, i)]);
    cst[m] = -0.5 * (D * log(2.0 * KWY_PI)) + ld + log(weights[m]);
  }
}

// weighted log prob wlp[t][m] = cst[m] - 0.5 * || Z_m (x_t - mu_m) ||^2
__global__ __launch_bounds__(FIT_NT) void k_fit_logprob(const double *__restrict__ X, int64_t n, int D, int M,
                                                       const double *__restrict__ means,
                                                       const double *__restrict__ Zp,
                                                       const double *__restrict__ cst,
                                                       double *__restrict__ wlp) {
  extern __shared__ double sm[];
  const int nt = (int)fit_tri(D), DP = D + 1;
  double *Z = sm;                     // packed
  double *dt = Z + nt;                // FIT_TILE x DP
  double *qp = dt + FIT_TILE * DP;    // 8 x FIT_TILE
  const int tid = threadIdx.x, m = blockIdx.x;
  const int64_t t0 = (int64_t)blockIdx.y * FIT_TILE;
  const double *zp = Zp + (size_t)m * nt, *mu = means + (size_t)m * D;
  for (int e = tid; e < nt; e += FIT_NT) Z[e] = zp[e];
  for (int e = tid; e < FIT_TILE * D; e += FIT_NT) {
    int tl = e / D, i = e % D;
    int64_t t = t0 + tl;
    dt[tl * DP + i] = t < n ? X[t * D + i] - mu[i] : 0.0;
  }
  __syncthreads();
  const int tl = tid & 63, jg = tid >> 6;  // 8 j-groups
  const int per = (D + 7) / 8;
  const int j0 = jg * per, j1 = min(D, j0 + per);
  const double *drow = dt + tl * DP;
  double q = 0.0;
  for (int j = j0; j < j1; ++j) {
    const double *zr = Z + tri(j, 0);
    double y = 0.0;
    for (int i = 0; i <= j; ++i) y += drow[i] * zr[i];
    q += y * y;
  }
  qp[jg * FIT_TILE + tl] = q;
  __syncthreads();
  if (tid < FIT_TILE) {
    int64_t t = t0 + tid;
    if (t < n) {
      double qq = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) qq += qp[g * FIT_TILE + tid];
      wlp[t * M + m] = cst[m] - 0.5 * qq;
    }
  }
}

// per frame: log-sum-exp over the mixtures; wlp is overwritten by the responsibilities;
// per-block sums of the frame log-likelihoods go to ll_part[blockIdx.x]
__global__ __launch_bounds__(KWY_THREADS) void k_fit_resp(double *__restrict__ wlp, int64_t n, int M,
                                                         double *__restrict__ ll_part) {
  __shared__ double red[8];
  const int64_t t = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x;
  double lse = 0.0;
  if (t < n) {
    double *w = wlp + t * M;
    double mx = -INFINITY;
    for (int m = 0; m < M; ++m) mx = fmax(mx, w[m]);
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += exp(w[m] - mx);
    lse = mx + log(s);
    for (int m = 0; m < M; ++m) w[m] = exp(w[m] - lse);
  }
  const double tot = kwy_block_sum(lse, red);
  if (threadIdx.x == 0) ll_part[blockIdx.x] = tot;
}

// local statistics: part[chunk][m][0] = sum_t r, part[chunk][m][1+i] = sum_t r x_i over the chunk's rows
__global__ __launch_bounds__(KWY_THREADS) void k_fit_sums(const double *__restrict__ X,
                                                         const double *__restrict__ resp, int64_t n, int D,
                                                         int M, int rows_per_chunk, double *__restrict__ part) {
  extern __shared__ double sm[];
  double *xs = sm;                        // FIT_SUM_ROWS x D
  double *rs = xs + FIT_SUM_ROWS * D;     // FIT_SUM_ROWS x M
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = min(n, r0 + rows_per_chunk);
  // thread -> (mixture m = tid % Mp, slice g = tid / Mp of the D+1 outputs), Mp = M rounded up to 2^k
  int Mp = 1;
  while (Mp < M) Mp <<= 1;
  const int groups = KWY_THREADS / Mp;
  const int m = tid % Mp, g = tid / Mp;
  const bool live = m < M;
  const int per = (D + 1 + groups - 1) / groups;  // outputs per thread (index 0 = nk, 1+i = sx_i)
  double acc[40];
#pragma unroll
  for (int q = 0; q < 40; ++q) acc[q] = 0.0;
  for (int64_t b = r0; b < r1; b += FIT_SUM_ROWS) {
    const int nr = (int)min((int64_t)FIT_SUM_ROWS, r1 - b);
    __syncthreads();
    for (int e = tid; e < nr * D; e += KWY_THREADS) xs[e] = X[b * D + e];
    for (int e = tid; e < nr * M; e += KWY_THREADS) rs[e] = resp[b * M + e];
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      const double rr = live ? rs[r * M + m] : 0.0;
      const double *xr = xs + r * D;
#pragma unroll
      for (int q = 0; q < 40; ++q) {
        const int o = g * per + q;
        if (q < per && o <= D) acc[q] += (o == 0) ? rr : rr * xr[o - 1];
      }
    }
  }
  if (!live) return;
  double *out = part + ((size_t)blockIdx.x * M + m) * (D + 1);
#pragma unroll
  for (int q = 0; q < 40; ++q) {
    const int o = g * per + q;
    if (q < per && o <= D) out[o] = acc[q];
  }
}

// out[e] = sum_c part[c][e]
__global__ void k_fit_reduce(const double *__restrict__ part, int nchunks, int64_t len, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(size_t)c * len + e];
  out[e] = s;
}

// local statistics: cpart[split][m] = sum over the split's rows of r (x - mu_m)(x - mu_m)'  (full D x D)
// 256 threads as a 16 x 16 grid, each owning a TS x TS tile of the D x D result (D <= 16*TS).
template <int TS>
__global__ __launch_bounds__(KWY_THREADS) void k_fit_cov(const double *__restrict__ X,
                                                        const double *__restrict__ resp, int64_t n, int D,
                                                        int M, const double *__restrict__ means,
                                                        double *__restrict__ cpart) {
  extern __shared__ double sm[];
  const int DP = 16 * TS;
  double *ds = sm;                       // FIT_COV_ROWS x DP (zero padded)
  double *rs = ds + FIT_COV_ROWS * DP;   // FIT_COV_ROWS
  const int tid = threadIdx.x, m = blockIdx.x, split = blockIdx.y;
  const int ti = tid >> 4, tj = tid & 15;
  const int64_t rows = (n + FIT_COV_SPLIT - 1) / FIT_COV_SPLIT;
  const int64_t r0 = split * rows, r1 = min(n, r0 + rows);
  const double *mu = means + (size_t)m * D;
  double acc[TS][TS];
#pragma unroll
  for (int a = 0; a < TS; ++a)
#pragma unroll
    for (int b = 0; b < TS; ++b) acc[a][b] = 0.0;
  for (int64_t b0 = r0; b0 < r1; b0 += FIT_COV_ROWS) {
    const int nr = (int)min((int64_t)FIT_COV_ROWS, r1 - b0);
    __syncthreads();
    for (int e = tid; e < FIT_COV_ROWS * DP; e += KWY_THREADS) {
      int r = e / DP, i = e % DP;
      ds[e] = (r < nr && i < D) ? X[(b0 + r) * D + i] - mu[i] : 0.0;
    }
    for (int r = tid; r < FIT_COV_ROWS; r += KWY_THREADS) rs[r] = r < nr ? resp[(b0 + r) * M + m] : 0.0;
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      const double rr = rs[r];
      const double *dr = ds + r * DP;
      double av[TS], bv[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) { av[a] = rr * dr[ti * TS + a]; bv[a] = dr[tj * TS + a]; }
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) acc[a][b] += av[a] * bv[b];
    }
  }
  double *out = cpart + ((size_t)split * M + m) * D * D;
#pragma unroll
  for (int a = 0; a < TS; ++a)
#pragma unroll
    for (int b = 0; b < TS; ++b) {
      int i = ti * TS + a, j = tj * TS + b;
      if (i < D && j < D) out[(size_t)i * D + j] = acc[a][b];
    }
}

// means = sx / nk   (nk already holds sum r + 10 eps)
__global__ void k_fit_means(const double *__restrict__ stats, int D, int M, double *__restrict__ means) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * D) return;
  const int m = e / D, i = e % D;
  const double nk = stats[(size_t)m * (D + 1)] + 10 * 2.220446049250313e-16;
  means[e] = stats[(size_t)m * (D + 1) + 1 + i] / nk;
}

// weights = nk / sum nk ; covs = sxx / nk + reg on the diagonal
__global__ void k_fit_finalize(const double *__restrict__ stats, const double *__restrict__ sxx, int D, int M,
                               double reg, double *__restrict__ weights, double *__restrict__ covs) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < M) {
    double tot = 0.0;
    for (int k = 0; k < M; ++k) tot += stats[(size_t)k * (D + 1)] + 10 * 2.220446049250313e-16;
    weights[e] = (stats[(size_t)e * (D + 1)] + 10 * 2.220446049250313e-16) / tot;
  }
  if (e >= (int64_t)M * D * D) return;
  const int m = (int)(e / ((int64_t)D * D));
  const int rem = (int)(e % ((int64_t)D * D));
  const int i = rem / D, j = rem % D;
  const double nk = stats[(size_t)m * (D + 1)] + 10 * 2.220446049250313e-16;
  double v = sxx[e] / nk;
  if (i == j) v += reg;
  covs[e] = v;
}

// ---- C ABI ------------------------------------------------------------------------------------------
static int fit_check(kwy_ctx *ctx, int64_t n, int D, int M) {
  if (!ctx) return KWY_EINVAL;
  if (n <= 0 || D <= 0 || D > 160 || M <= 0 || M > 256) {
    ctx->err = "gmm_em: need 0 < D <= 160 and 0 < M <= 256";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_gmm_em_scratch_bytes(int64_t n, int D, int M, int64_t *bytes) {
  if (!bytes) return KWY_EINVAL;
  const int64_t nchunks = (n + 4095) / 4096;
  *bytes = (int64_t)(kwy_pad(sizeof(double) * fit_tri(D) * M) + kwy_pad(sizeof(double) * (size_t)M * D * D) +
                     kwy_pad(sizeof(double) * M) + kwy_pad(sizeof(double) * (size_t)nchunks * M * (D + 1)) +
                     kwy_pad(sizeof(double) * (size_t)FIT_COV_SPLIT * M * D * D) + kwy_pad(64) + 4096);
  return KWY_OK;
}

// E-step on the local shard.  resp: n x M (out).  loglik_parts: ceil(n/256) doubles (out): their sum is
// sum_t log p(x_t).  status_out (device int): nonzero if a covariance was not positive definite.
extern "C" int kwy_gmm_em_estep_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M,
                                    const double *weights, const double *means, const double *covs,
                                    double *resp, double *loglik_parts, int *status_out) {
  KWY_TRY(fit_check(ctx, n, D, M));
  if (!X || !weights || !means || !covs || !resp || !loglik_parts || !status_out) { ctx->err = "gmm_em_estep: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const size_t nt = fit_tri(D);
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * nt * M) + kwy_pad(sizeof(double) * (size_t)M * D * D) +
                                   kwy_pad(sizeof(double) * M)));
  double *Zp = kwy_arena<double>(ctx, nt * M);
  double *zcol = kwy_arena<double>(ctx, (size_t)M * D * D);
  double *cst = kwy_arena<double>(ctx, M);
  if (!Zp || !zcol || !cst) { ctx->err = "gmm_em_estep: scratch"; return KWY_ENOMEM; }
  KWY_HIP(hipMemsetAsync(status_out, 0, sizeof(int), ctx->stream));
  const size_t lds_prec = sizeof(double) * nt;
  const size_t lds_lp = sizeof(double) * (nt + (size_t)FIT_TILE * (D + 1) + 8 * FIT_TILE);
  if (lds_lp > 160 * 1024) { ctx->err = "gmm_em_estep: feature dimension too large for LDS"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_prec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prec));
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_logprob, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lp));
  hipLaunchKernelGGL(k_fit_prec, dim3(M), dim3(KWY_THREADS), lds_prec, ctx->stream, weights, covs, D, Zp, zcol, cst,
                     status_out);
  KWY_PROF(ctx, "k_fit_logprob", hipLaunchKernelGGL(k_fit_logprob, dim3(M, (unsigned)((n + FIT_TILE - 1) / FIT_TILE)), dim3(FIT_NT), lds_lp,
                     ctx->stream, X, n, D, M, means, Zp, cst, resp));
  hipLaunchKernelGGL(k_fit_resp, dim3((unsigned)((n + KWY_THREADS - 1) / KWY_THREADS)), dim3(KWY_THREADS), 0,
                     ctx->stream, resp, n, M, loglik_parts);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// stats[m][0] = sum_t r[t][m], stats[m][1+i] = sum_t r[t][m] x[t][i]   (local shard)
extern "C" int kwy_gmm_em_sums_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                                   double *stats) {
  KWY_TRY(fit_check(ctx, n, D, M));
  if (!X || !resp || !stats) { ctx->err = "gmm_em_sums: null pointer"; return KWY_EINVAL; }
  {
    int Mp = 1;
    while (Mp < M) Mp <<= 1;
    const int groups = 256 / Mp;
    if ((D + 1 + groups - 1) / groups > 40) { ctx->err = "gmm_em_sums: D too large for this component count"; return KWY_EINVAL; }
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int rows_per_chunk = 4096;
  const int nchunks = (int)((n + rows_per_chunk - 1) / rows_per_chunk);
  const int64_t len = (int64_t)M * (D + 1);
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)nchunks * len)));
  double *part = kwy_arena<double>(ctx, (size_t)nchunks * len);
  if (!part) { ctx->err = "gmm_em_sums: scratch"; return KWY_ENOMEM; }
  const size_t lds = sizeof(double) * ((size_t)FIT_SUM_ROWS * D + (size_t)FIT_SUM_ROWS * M);
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_sums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_fit_sums, dim3(nchunks), dim3(KWY_THREADS), lds, ctx->stream, X, resp, n, D, M, rows_per_chunk,
                     part);
  hipLaunchKernelGGL(k_fit_reduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, part, nchunks, len,
                     stats);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// means[m][i] = stats[m][1+i] / (stats[m][0] + 10 eps)      (stats: globally reduced)
extern "C" int kwy_gmm_em_means_dev(kwy_ctx *ctx, const double *stats, int D, int M, double *means) {
  KWY_TRY(fit_check(ctx, 1, D, M));
  if (!stats || !means) { ctx->err = "gmm_em_means: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_fit_means, dim3((M * D + 255) / 256), dim3(256), 0, ctx->stream, stats, D, M, means);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// sxx[m] = sum_t r[t][m] (x_t - mu_m)(x_t - mu_m)'    (local shard; full D x D per mixture)
extern "C" int kwy_gmm_em_cov_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                                  const double *means, double *sxx) {
  KWY_TRY(fit_check(ctx, n, D, M));
  if (!X || !resp || !means || !sxx) { ctx->err = "gmm_em_cov: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int64_t len = (int64_t)M * D * D;
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)FIT_COV_SPLIT * len)));
  double *cpart = kwy_arena<double>(ctx, (size_t)FIT_COV_SPLIT * len);
  if (!cpart) { ctx->err = "gmm_em_cov: scratch"; return KWY_ENOMEM; }
  const int TS = (D + 15) / 16;
  const size_t lds = sizeof(double) * ((size_t)FIT_COV_ROWS * 16 * TS + FIT_COV_ROWS);
#define LAUNCH_COV(ts)                                                                                              \
  do {                                                                                                              \
    KWY_HIP(hipFuncSetAttribute((const void *)k_fit_cov<ts>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    KWY_PROF(ctx, "k_fit_cov", hipLaunchKernelGGL(k_fit_cov<ts>, dim3(M, FIT_COV_SPLIT), dim3(KWY_THREADS), lds, ctx->stream, X, resp, n, D, M, means, cpart)); \
  } while (0)
  switch (TS) {
    case 1: LAUNCH_COV(1); break;
    case 2: LAUNCH_COV(2); break;
    case 3: LAUNCH_COV(3); break;
    case 4: LAUNCH_COV(4); break;
    case 5: LAUNCH_COV(5); break;
    case 6: LAUNCH_COV(6); break;
    case 7: LAUNCH_COV(7); break;
    case 8: LAUNCH_COV(8); break;
    case 9: LAUNCH_COV(9); break;
    default: LAUNCH_COV(10); break;
  }
#undef LAUNCH_COV
  hipLaunchKernelGGL(k_fit_reduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, cpart,
                     FIT_COV_SPLIT, len, sxx);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// weights = nk / sum nk;  covs = sxx / nk + reg_covar * I       (stats, sxx: globally reduced)
extern "C" int kwy_gmm_em_finalize_dev(kwy_ctx *ctx, const double *stats, const double *sxx, int D, int M,
                                       double reg_covar, double *weights, double *covs) {
  KWY_TRY(fit_check(ctx, 1, D, M));
  if (!stats || !sxx || !weights || !covs) { ctx->err = "gmm_em_finalize: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int64_t len = (int64_t)M * D * D;
  hipLaunchKernelGGL(k_fit_finalize, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, stats, sxx, D, M,
                     reg_covar, weights, covs);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}
