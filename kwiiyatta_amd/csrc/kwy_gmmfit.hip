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
// The n x D by D x D (lower-triangular) product runs on v_mfma_f64_16x16x4_f64.  A workgroup owns one
// mixture: its packed triangular Z stays in LDS (83.5 KB for D = 144) while the workgroup walks over
// frame tiles; wavefront w stages and multiplies frames 16w..16w+15 of a tile on its own (no block
// barrier inside the walk), fetching the next tile's rows into registers while the MFMAs of the
// current one run.  Per 16-column block of Z only the k-steps up to the block's diagonal are issued
// (180 instead of 324 MFMAs per wavefront and tile for D = 144).
// Lane map: A[row l&15][k l>>4], B[k l>>4][col l&15], D[row (l>>4)+4r][col l&15].
typedef double fit_v4f64 __attribute__((ext_vector_type(4)));
#define FIT_LP_NT 256
#define FIT_LP_COLS 3   // feature columns per lane when staging a row: D <= 192

__global__ __launch_bounds__(FIT_LP_NT) void k_fit_logprob(const double *__restrict__ X, int64_t n, int D, int M,
                                                          const double *__restrict__ means,
                                                          const double *__restrict__ Zp,
                                                          const double *__restrict__ cst,
                                                          double *__restrict__ wlp) {
  extern __shared__ double sm[];
  const int nt = (int)fit_tri(D), Kp = (D + 3) & ~3, NP = (D + 15) & ~15, ZS = Kp + 1;
  double *Z = sm;                 // packed lower triangle
  double *dts = Z + nt;           // FIT_TILE x ZS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.x;
  const double *zp = Zp + (size_t)m * nt, *mu = means + (size_t)m * D;
  for (int e = tid; e < nt; e += FIT_LP_NT) Z[e] = zp[e];
  const double cm = cst[m];
  double muv[FIT_LP_COLS];
#pragma unroll
  for (int c = 0; c < FIT_LP_COLS; ++c) muv[c] = (lane + 64 * c < D) ? mu[lane + 64 * c] : 0.0;
  const int ar = lane & 15, ak = lane >> 4;
  const int64_t ntiles = (n + FIT_TILE - 1) / FIT_TILE;
  double *myrows = dts + (16 * wv) * ZS;
  // rows of the first tile
  double xv[16][FIT_LP_COLS];
  auto fetch = [&](int64_t tile) {
    const int64_t tb = tile * FIT_TILE + 16 * wv;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        xv[r][c] = (tile < ntiles && tb + r < n && i < D) ? X[(tb + r) * D + i] : muv[c];
      }
  };
  fetch(blockIdx.y);
  __syncthreads();  // Z complete
  for (int64_t tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        if (i < Kp) myrows[r * ZS + i] = xv[r][c] - muv[c];
      }
    __builtin_amdgcn_wave_barrier();
    fetch(tile + gridDim.y);  // in flight during the MFMAs below
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    const double *arow = myrows + ar * ZS + ak;
    for (int nb = 0; nb < NP / 16; ++nb) {
      const int j = 16 * nb + ar;
      const double *zrow = Z + tri(j < D ? j : 0, 0) + ak;
      fit_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int ks = 0; ks < 4 * nb; ++ks)  // strictly below the diagonal block: no masking (j >= 16 nb > i)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * ks], j < D ? zrow[4 * ks] : 0.0, acc, 0, 0, 0);
#pragma unroll
      for (int dgn = 0; dgn < 4; ++dgn) {  // the diagonal block
        const int ks = 4 * nb + dgn;
        if (4 * ks < Kp) {
          const int i = 4 * ks + ak;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * ks], (j < D && i <= j) ? zrow[4 * ks] : 0.0, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] += acc[r] * acc[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = q[r];
      v += kwy_dpp_f64<0x111>(v);
      v += kwy_dpp_f64<0x112>(v);
      v += kwy_dpp_f64<0x114>(v);
      v += kwy_dpp_f64<0x118>(v);
      q[r] = v;
    }
    if (ar == 15) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t t = tile * FIT_TILE + 16 * wv + ak + 4 * r;
        if (t < n) wlp[t * M + m] = cm - 0.5 * q[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
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
// A D x n by n x D product with the frame index as the MFMA's k: v_mfma_f64_16x16x4_f64 consumes four
// frames per instruction, A = r_t d_t[i], B = d_t[j].  A workgroup owns (mixture, row split) and
// walks over 64-frame tiles staged in LDS; wavefront w accumulates the 16-row blocks w, w+4, w+8 of
// the result against all 16-column blocks (the full square: the symmetric half would need a
// per-wavefront tile list, i.e. dynamically indexed accumulators).
#define FIT_COV_NB 10      // 16-column blocks: D <= 160
#define FIT_COV_RB 3       // 16-row blocks per wavefront: ceil(10 / 4)
__global__ __launch_bounds__(KWY_THREADS) void k_fit_cov(const double *__restrict__ X,
                                                        const double *__restrict__ resp, int64_t n, int D,
                                                        int M, const double *__restrict__ means,
                                                        double *__restrict__ cpart) {
  extern __shared__ double sm[];
  const int NP = (D + 15) & ~15, nb = NP / 16, ZS = NP + 1;
  double *ds = sm;                 // FIT_TILE x ZS, zero padded columns
  double *rs = ds + FIT_TILE * ZS; // FIT_TILE
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.x, split = blockIdx.y;
  const int ar = lane & 15, ak = lane >> 4;
  const int64_t rows = (n + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = split * rows, r1 = min(n, r0 + rows);
  const double *mu = means + (size_t)m * D;
  double muv[FIT_LP_COLS];
#pragma unroll
  for (int c = 0; c < FIT_LP_COLS; ++c) muv[c] = (lane + 64 * c < D) ? mu[lane + 64 * c] : 0.0;
  fit_v4f64 acc[FIT_COV_RB][FIT_COV_NB];
#pragma unroll
  for (int a = 0; a < FIT_COV_RB; ++a)
#pragma unroll
    for (int b = 0; b < FIT_COV_NB; ++b) acc[a][b] = fit_v4f64{0.0, 0.0, 0.0, 0.0};
  // this wavefront stages rows 16w..16w+15 of every tile: fetched one tile ahead into registers
  double xv[16][FIT_LP_COLS], rv = 0.0;
  auto fetch = [&](int64_t b0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        const int64_t t = b0 + 16 * wv + r;
        xv[r][c] = (t < r1 && i < D) ? X[t * D + i] : muv[c];
      }
    const int64_t t = b0 + 16 * wv + lane;
    rv = (lane < 16 && t < r1) ? resp[t * M + m] : 0.0;
  };
  fetch(r0);
  for (int64_t b0 = r0; b0 < r1; b0 += FIT_TILE) {
    __syncthreads();  // the previous tile has been consumed
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        if (i < NP) ds[(16 * wv + r) * ZS + i] = xv[r][c] - muv[c];
      }
    if (lane < 16) rs[16 * wv + lane] = rv;
    __syncthreads();
    fetch(b0 + FIT_TILE);
#pragma unroll 2
    for (int ks = 0; ks < FIT_TILE / 4; ++ks) {
      const double *drow = ds + (4 * ks + ak) * ZS + ar;
      const double rr = rs[4 * ks + ak];
      double bv[FIT_COV_NB], av[FIT_COV_RB];
#pragma unroll
      for (int b = 0; b < FIT_COV_NB; ++b) bv[b] = b < nb ? drow[16 * b] : 0.0;
#pragma unroll
      for (int a = 0; a < FIT_COV_RB; ++a) av[a] = (wv + 4 * a < nb) ? rr * drow[16 * (wv + 4 * a)] : 0.0;
#pragma unroll
      for (int a = 0; a < FIT_COV_RB; ++a) {
        if (wv + 4 * a < nb) {
#pragma unroll
          for (int b = 0; b < FIT_COV_NB; ++b)
            if (b < nb) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
  double *out = cpart + ((size_t)split * M + m) * D * D;
#pragma unroll
  for (int a = 0; a < FIT_COV_RB; ++a)
#pragma unroll
    for (int b = 0; b < FIT_COV_NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * (wv + 4 * a) + ak + 4 * r, j = 16 * b + ar;
        if (wv + 4 * a < nb && b < nb && i < D && j < D) out[(size_t)i * D + j] = acc[a][b][r];
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

// frame-tile walkers per mixture: one workgroup fits per CU
static unsigned fit_lp_splits(int64_t n, int M) {
  const int64_t ntiles = (n + FIT_TILE - 1) / FIT_TILE;
  int64_t s = (256 + M - 1) / M;
  if (s > ntiles) s = ntiles;
  return (unsigned)(s < 1 ? 1 : s);
}

// ---- C ABI ------------------------------------------------------------------------------------------
static int fit_check(kwy_ctx *ctx, int64_t n, int D, int M) {
  if (!ctx) return KWY_EINVAL;
  if (n <= 0 || D <= 0 || D > 160 || D > 64 * FIT_LP_COLS || M <= 0 || M > 256) {
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
  const size_t lds_lp = sizeof(double) * (nt + (size_t)FIT_TILE * (((D + 3) & ~3) + 1));
  if (lds_lp > 160 * 1024) { ctx->err = "gmm_em_estep: feature dimension too large for LDS"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_prec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prec));
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_logprob, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lp));
  hipLaunchKernelGGL(k_fit_prec, dim3(M), dim3(KWY_THREADS), lds_prec, ctx->stream, weights, covs, D, Zp, zcol, cst,
                     status_out);
  KWY_PROF(ctx, "k_fit_logprob", hipLaunchKernelGGL(k_fit_logprob, dim3(M, fit_lp_splits(n, M)), dim3(FIT_LP_NT), lds_lp,
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
  const int NP = (D + 15) & ~15;
  const size_t lds = sizeof(double) * ((size_t)FIT_TILE * (NP + 1) + FIT_TILE);
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_cov, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_fit_cov", hipLaunchKernelGGL(k_fit_cov, dim3(M, FIT_COV_SPLIT), dim3(KWY_THREADS), lds, ctx->stream, X, resp, n, D, M, means, cpart));
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
