// kwy_mlpg.hip -- GMM-based spectral mapping with MLPG on gfx950.
//
// Replaces, for the converter-apply step of the reference
// (kwiiyatta/converter/delta.py:39-50 and gmm.py:28-34):
//     X  = nnmnkwii.preprocessing.delta_features(x, DELTA_WINDOWS)
//     y  = nnmnkwii.baseline.gmm.MLPG(gmm, windows=DELTA_WINDOWS, diff).transform(X)[:, :d]
//
// Kernels:
//   k_gmm_prep   one workgroup per mixture: Cholesky of S_xx in LDS, its inverse,
//                A = S_yx S_xx^-1, diagonal conditional variance, log-normaliser
//   k_delta      static + delta + delta-delta features (3-tap correlations)
//   k_gmm_logp   (frame tile x mixture) workgroups: ||L^-1 (x - mu)||^2 from LDS
//   k_gmm_cond   per frame: arg-max mixture, conditional mean E and variance D
//   k_mlpg_build per (frame, dim): the pentadiagonal normal equations W'PW, W'P mu
//   k_mlpg_chunks / k_mlpg_finish  per static dim: banded Cholesky, partitioned into chunks (nested dissection)
//
// Algorithmic HBM bytes per frame: d*8 in, d*8 out (+ the GMM once per call).
#include <math.h>

#include <algorithm>

#include "kwy_internal.hpp"

#define ML_TILE 64  // frames per k_gmm_logp workgroup

struct ml_dims { int d, D, M; int64_t T; };

// A conversion call handles a batch of up to KWY_BATCH_MAX utterances (a single call is a batch of one).  The frames of
// all of them form ONE matrix for the frame-parallel kernels (log-densities, conditional means: dm.T = all frames);
// the kernels that look at neighbouring frames (deltas, normal equations) and the trajectory solves find their
// utterance u by start[u] <= frame < start[u + 1].  Descriptors travel by value in the kernel arguments.
struct ml_part { int P, base, extra; };
struct ml_seg {
  const double *x;          // T x d static features, rows ldx doubles apart
  double *y;                // T x d converted, rows ldy apart
  const double *keep_in;    // (may be null) one more column copied through, ldx / ldy apart
  double *keep_out;
  ml_part pt;               // chunks of the partitioned trajectory solve
};
struct ml_batch {
  int n, ldx, ldy;
  int64_t start[KWY_BATCH_MAX + 1];
  ml_seg u[KWY_BATCH_MAX];
  __device__ __forceinline__ int find(int64_t t) const {
    int k = 0;
    while (k + 1 < n && t >= start[k + 1]) ++k;
    return k;
  }
};

// ---- per-mixture preparation -----------------------------------------------------
// layout of the prepared model (doubles), per mixture m:
//   Z   [D*D]  lower-triangular inverse of chol(S_xx)   (row j: Z[j][0..j])
//   A   [D*D]  S_yx S_xx^-1
//   mux [D], muy [D], dvar [D], cst [1] (+ padding)
__host__ __device__ static inline size_t ml_model_stride(int D) { return (size_t)2 * D * D + 3 * D + 8; }

__global__ __launch_bounds__(KWY_THREADS) void k_gmm_prep(const double *__restrict__ weights,
                                                         const double *__restrict__ means,
                                                         const double *__restrict__ covs, int D, int diff,
                                                         double *__restrict__ model, int *__restrict__ status) {
  extern __shared__ double smem[];
  double *L = smem;          // D x D
  double *Z = L + D * D;     // D x D
  double *W = Z + D * D;     // D x D
  const int tid = threadIdx.x, m = blockIdx.x, D2 = 2 * D;
  const double *C = covs + (size_t)m * D2 * D2;
  double *out = model + (size_t)m * ml_model_stride(D);
  double *oZ = out, *oA = oZ + D * D, *omux = oA + D * D, *omuy = omux + D, *odv = omuy + D, *ocst = odv + D;

  auto Sxx = [&](int i, int j) { return C[(size_t)i * D2 + j]; };
  auto Sxy = [&](int i, int j) { return diff ? C[(size_t)i * D2 + D + j] - C[(size_t)i * D2 + j] : C[(size_t)i * D2 + D + j]; };
  auto Syx = [&](int i, int j) { return diff ? Sxy(j, i) : C[(size_t)(D + i) * D2 + j]; };
  auto Syy = [&](int i, int j) {
    double v = C[(size_t)(D + i) * D2 + D + j];
    if (diff) v = C[(size_t)i * D2 + j] + v - C[(size_t)i * D2 + D + j] - C[(size_t)(D + i) * D2 + j];
    return v;
  };

  for (int e = tid; e < D * D; e += KWY_THREADS) L[e] = Sxx(e / D, e % D);
  __syncthreads();
  // right-looking Cholesky (same subtraction order per entry as the textbook left-looking form)
  for (int j = 0; j < D; ++j) {
    const double piv = L[j * D + j];
    if (!(piv > 0.0)) { if (tid == 0) atomicExch(status, 1); return; }
    const double dj = sqrt(piv);
    __syncthreads();
    if (tid == 0) L[j * D + j] = dj;
    for (int i = j + 1 + tid; i < D; i += KWY_THREADS) L[i * D + j] = L[i * D + j] / dj;
    __syncthreads();
    const int rem = D - j - 1;
    for (int e = tid; e < rem * rem; e += KWY_THREADS) {
      int i = j + 1 + e / rem, k = j + 1 + e % rem;
      if (k <= i) L[i * D + k] -= L[i * D + j] * L[k * D + j];
    }
    __syncthreads();
  }
  // Z = L^-1 (lower): column c by thread c
  for (int c = tid; c < D; c += KWY_THREADS) {
    for (int i = 0; i < D; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) v -= L[i * D + k] * Z[k * D + c];
      Z[i * D + c] = (i < c) ? 0.0 : v / L[i * D + i];
    }
  }
  __syncthreads();
  // W = S_yx Z'   (W[r][j] = sum_i Syx[r][i] Z[j][i])
  for (int e = tid; e < D * D; e += KWY_THREADS) {
    int r = e / D, j = e % D;
    double v = 0.0;
    for (int i = 0; i <= j; ++i) v += Syx(r, i) * Z[j * D + i];
    W[e] = v;
  }
  __syncthreads();
  // A = W Z       (A[r][c] = sum_j W[r][j] Z[j][c])
  for (int e = tid; e < D * D; e += KWY_THREADS) {
    int r = e / D, c = e % D;
    double v = 0.0;
    for (int j = c; j < D; ++j) v += W[r * D + j] * Z[j * D + c];
    oA[e] = v;
    oZ[e] = Z[e];
  }
  for (int i = tid; i < D; i += KWY_THREADS) {
    double mx = means[(size_t)m * D2 + i], my = means[(size_t)m * D2 + D + i];
    omux[i] = mx;
    omuy[i] = diff ? my - mx : my;
    odv[i] = Syy(i, i) - Syx(i, i) / Sxx(i, i) * Sxy(i, i);
  }
  if (tid == 0) {
    double ld = 0.0;
    for (int i = 0; i < D; ++i) ld -= log(L[i * D + i]);
    ocst[0] = -0.5 * (D * log(2.0 * KWY_PI)) + ld + log(weights[m]);
  }
}

// ---- delta features -----------------------------------------------------------------
// DELTA_WINDOWS (kwiiyatta/converter/delta.py:8-12): [1], [-0.5, 0, 0.5], [1, -2, 1]
// x rows are ldx doubles apart.  keep_in / keep_out (may be null): one more column that is copied through
// unchanged (the power coefficient c0 of a mel-cepstrum, kwiiyatta/converter/mcep.py:57-59), ldx / ldy apart.
__global__ void k_delta(ml_batch b, ml_dims dm, double *__restrict__ X) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dm.T * dm.d) return;
  const int64_t tg = e / dm.d;              // frame of the batch
  const int c = (int)(e % dm.d);
  const int u = b.find(tg);
  const int64_t t = tg - b.start[u], T = b.start[u + 1] - b.start[u];
  const double *__restrict__ x = b.u[u].x;
  const int ldx = b.ldx, ldy = b.ldy;
  if (b.u[u].keep_out && c == 0) b.u[u].keep_out[t * ldy] = b.u[u].keep_in[t * ldx];
  const double xm = t > 0 ? x[(t - 1) * ldx + c] : 0.0;
  const double x0 = x[t * ldx + c];
  const double xp = t + 1 < T ? x[(t + 1) * ldx + c] : 0.0;
  double *o = X + tg * dm.D;
  o[c] = 0.0 + x0 * 1.0;
  double s = 0.0;
  if (t > 0) s += xm * -0.5;
  s += x0 * 0.0;
  if (t + 1 < T) s += xp * 0.5;
  o[dm.d + c] = s;
  s = 0.0;
  if (t > 0) s += xm * 1.0;
  s += x0 * -2.0;
  if (t + 1 < T) s += xp * 1.0;
  o[2 * dm.d + c] = s;
}

// ---- log p(x_t | m) up to the shared constant -------------------------------------------
// || Z_m (x_t - mu_m) ||^2 for a tile of 64 frames and one mixture is a 64 x D times D x D
// (lower-triangular) product: the one MFMA-shaped piece of the conversion path.  One workgroup
// keeps Z_m (zero above the diagonal, rows padded) in LDS and walks over frame tiles; wavefront w
// owns frames 16w..16w+15 and, per 16-column block of Z, runs v_mfma_f64_16x16x4_f64 over the
// k-steps that reach the block's diagonal (58 instead of 90 MFMAs per wavefront and tile for D = 72).
// Lane map of that instruction: A[row l&15][k l>>4], B[k l>>4][col l&15], D[row (l>>4)+4r][col l&15].
typedef double ml_v4f64 __attribute__((ext_vector_type(4)));

__host__ __device__ static inline int ml_kp(int D) { return (D + 3) & ~3; }
__host__ __device__ static inline int ml_np(int D) { return (D + 15) & ~15; }

// NW wavefronts per workgroup, 16 frames each (round 4: eight where the LDS allows -- a workgroup fills a CU alone, so
// its wavefronts are all the latency hiding there is: 0.68 -> see DESIGN.md section 5 for a wave of 35 216 frames)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_gmm_logp(const double *__restrict__ X, ml_dims dm,
                                                         const double *__restrict__ model,
                                                         double *__restrict__ logp) {
  extern __shared__ double smem[];
  const int D = dm.D, Kp = ml_kp(D), NP = ml_np(D), ZS = Kp + 1;
  double *Zs = smem;             // NP x ZS
  constexpr int TILE = 16 * NW;
  double *dts = Zs + NP * ZS;    // TILE x ZS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.y;
  const double *mod = model + (size_t)m * ml_model_stride(D);
  const double *mZ = mod, *mux = mod + 2 * D * D;
  const double cst = mod[2 * D * D + 3 * D];
  for (int jb = 4 * wv; jb < NP; jb += 4 * NW) {  // four rows per wavefront and step, loads batched
    double z0[4], z1[4];
    const int i0 = lane, i1 = lane + 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = jb + r;
      z0[r] = (j < D && i0 <= j) ? mZ[j * D + i0] : 0.0;
      z1[r] = (j < D && i1 <= j) ? mZ[j * D + i1] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = jb + r;
      if (j < NP) {
        if (i0 < Kp) Zs[j * ZS + i0] = z0[r];
        if (i1 < Kp) Zs[j * ZS + i1] = z1[r];
        for (int i = lane + 128; i < Kp; i += 64) Zs[j * ZS + i] = (j < D && i <= j) ? mZ[j * D + i] : 0.0;
      }
    }
  }
  const int ar = lane & 15, ak = lane >> 4;
  const int64_t ntiles = (dm.T + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t t0 = tile * TILE;
    __syncthreads();  // the previous tile has been consumed (and Zs is complete)
    // every wavefront stages the 16 frames it multiplies itself
    {
      // all loads of the wavefront's 16 rows in flight before the first store (two columns per lane)
      const int i0 = lane, i1 = lane + 64;
      const double mu0 = i0 < D ? mux[i0] : 0.0, mu1 = i1 < D ? mux[i1] : 0.0;
      double v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t t = t0 + 16 * wv + r;
        v0[r] = (t < dm.T && i0 < D) ? X[t * D + i0] : mu0;
        v1[r] = (t < dm.T && i1 < D) ? X[t * D + i1] : mu1;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double *row = dts + (16 * wv + r) * ZS;
        if (i0 < Kp) row[i0] = v0[r] - mu0;
        if (i1 < Kp) row[i1] = v1[r] - mu1;
      }
      for (int i = lane + 128; i < Kp; i += 64)   // wider features: the remaining columns, plainly
        for (int r = 0; r < 16; ++r) {
          const int64_t t = t0 + 16 * wv + r;
          dts[(16 * wv + r) * ZS + i] = (t < dm.T && i < D) ? X[t * D + i] - mux[i] : 0.0;
        }
    }
    __syncthreads();
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    const double *arow = dts + (16 * wv + ar) * ZS + ak;
    for (int nt = 0; nt < NP / 16; ++nt) {
      ml_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
      const int ksteps = min(Kp / 4, 4 * (nt + 1));
      const double *brow = Zs + (16 * nt + ar) * ZS + ak;
      // (built with -mllvm -amdgpu-mfma-vgpr-form=1: the accumulator stays in VGPRs across the loop
      // instead of being copied to and from AGPRs around every MFMA)
#pragma unroll 4
      for (int ks = 0; ks < ksteps; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * ks], brow[4 * ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] += acc[r] * acc[r];
    }
    // sum over the 16 columns held by the lanes of one row group: lane 15 of the group ends up with the total
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
        const int64_t t = t0 + 16 * wv + ak + 4 * r;
        if (t < dm.T) logp[t * dm.M + m] = cst - 0.5 * q[r];
      }
    }
  }
}

// ---- arg-max mixture, conditional mean / variance --------------------------------------------
__global__ __launch_bounds__(128) void k_gmm_cond(const double *__restrict__ X, ml_dims dm,
                                                 const double *__restrict__ model,
                                                 const double *__restrict__ logp,
                                                 double *__restrict__ E, double *__restrict__ Dv,
                                                 int *__restrict__ mix) {
  __shared__ int s_m;
  __shared__ double dsh[256];
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, D = dm.D;
  if (tid == 0) {
    int bm = 0;
    double best = -INFINITY;
    for (int m = 0; m < dm.M; ++m) {
      double lp = logp[t * dm.M + m];
      if (lp > best) { best = lp; bm = m; }
    }
    s_m = bm;
    mix[t] = bm;
  }
  __syncthreads();
  const double *mod = model + (size_t)s_m * ml_model_stride(D);
  const double *A = mod + D * D, *mux = A + D * D, *muy = mux + D, *dvar = muy + D;
  for (int i = tid; i < D; i += blockDim.x) dsh[i] = X[t * D + i] - mux[i];
  __syncthreads();
  for (int i = tid; i < D; i += blockDim.x) {
    double v = muy[i];
    const double *ar = A + (size_t)i * D;
    for (int k = 0; k < D; ++k) v += ar[k] * dsh[k];
    E[t * D + i] = v;
    Dv[t * D + i] = dvar[i];
  }
}

// ---- frame-wise conversion (MLPG switched off): posterior-weighted conditional mean ------------------------
// nnmnkwii MLPGBase.transform, what GMMFeatureConverter.convert(mlpg=False) runs (kwiiyatta/converter/gmm.py:28-34
// with windows[0:1]): y_t = sum_m p(m | x_t) (mu_y,m + A_m (x_t - mu_x,m)), the posterior being the softmax of the
// weighted log-densities (sklearn predict_proba).  One workgroup per frame.
__global__ __launch_bounds__(128) void k_gmm_soft(const double *__restrict__ X, ml_dims dm,
                                                 const double *__restrict__ model,
                                                 const double *__restrict__ logp, double *__restrict__ Y) {
  __shared__ double post[256];
  __shared__ double xs[192];
  __shared__ double red[4];
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, D = dm.D, M = dm.M;
  double mx = -INFINITY;
  for (int m = tid; m < M; m += 128) mx = fmax(mx, logp[t * M + m]);
  mx = kwy_wave_max_f64(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  for (int i = tid; i < D; i += 128) xs[i] = X[t * D + i];
  __syncthreads();
  mx = fmax(red[0], red[1]);
  double se = 0.0;
  for (int m = tid; m < M; m += 128) { const double e = exp(logp[t * M + m] - mx); post[m] = e; se += e; }
  se = kwy_wave_sum(se);
  __syncthreads();
  if ((tid & 63) == 0) red[2 + (tid >> 6)] = se;
  __syncthreads();
  const double inv = 1.0 / (red[2] + red[3]);
  for (int i = tid; i < D; i += 128) {
    double y = 0.0;
    for (int m = 0; m < M; ++m) {
      const double pm = post[m] * inv;
      if (pm == 0.0) continue;
      const double *mod = model + (size_t)m * ml_model_stride(D);
      const double *ar = mod + D * D + (size_t)i * D, *mux = mod + 2 * D * D, *muy = mux + D;
      double v = muy[i];
      for (int k = 0; k < D; ++k) v += ar[k] * (xs[k] - mux[k]);
      y += pm * v;
    }
    Y[t * D + i] = y;
  }
}

// ---- MLPG: normal equations ---------------------------------------------------------------------
// rec[t][c][0..3] = P[t][t], P[t][t-1], P[t][t-2], rhs[t]: one 32-byte record per (row, dimension), so that the
// solver moves it with two 16-byte accesses per lane (its sweeps are bound by the vector-memory instruction rate of
// the one CU they run on)
__device__ __forceinline__ double ml_wcoef(int w, int k) {  // window w at offset k in [-1, 1]
  if (w == 0) return k == 0 ? 1.0 : 0.0;
  if (w == 1) return k == -1 ? -0.5 : (k == 0 ? 0.0 : 0.5);
  return k == 0 ? -2.0 : 1.0;
}

__global__ void k_mlpg_build(const double *__restrict__ E, const double *__restrict__ Dv, ml_batch bt, ml_dims dm,
                             double *__restrict__ rec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dm.T * dm.d) return;
  const int64_t ag = e / dm.d;
  const int c = (int)(e % dm.d);
  const int u = bt.find(ag);
  const int64_t t0 = bt.start[u], a = ag - t0, T = bt.start[u + 1] - t0;    // row a of utterance u's system
  E += t0 * dm.D;
  Dv += t0 * dm.D;
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, b = 0.0;
  for (int w = 0; w < 3; ++w) {
    const int l = w == 0 ? 0 : 1;
    for (int64_t t = a - l; t <= a + l; ++t) {  // frames whose window touches row a (ascending t)
      if (t < 0 || t >= T) continue;
      const int k1 = (int)(a - t);
      const double prec = 1 / Dv[t * dm.D + w * dm.d + c];
      const double bm = prec * E[t * dm.D + w * dm.d + c];
      const double ca = ml_wcoef(w, k1);
      b += ca * bm;
      for (int k2 = -l; k2 <= k1; ++k2) {
        const int64_t cc = t + k2;
        if (cc < 0 || cc >= T) continue;
        const double v = ca * prec * ml_wcoef(w, k2);
        const int off = (int)(a - cc);
        if (off == 0) p0 += v; else if (off == 1) p1 += v; else p2 += v;
      }
    }
  }
  double2 *o = (double2 *)rec + e * 2;
  o[0] = make_double2(p0, p1);
  o[1] = make_double2(p2, b);
}

// ---- MLPG: the T x T pentadiagonal SPD systems (one per static dimension), partitioned --------------------
// Round 1 walked the T rows serially, one lane per dimension: 2 x 2201 dependent steps of ~170 ns (0.76 ms, the
// longest kernel on the latency path of a pair).  Now the rows are cut into P chunks with a separator of two rows
// between neighbours (bandwidth 2: removing two rows decouples the chunks), and the elimination runs in
// nested-dissection order:
//   1. every (chunk, dimension) lane -- 64/d chunks per wavefront -- factors its chunk (banded Cholesky) and solves
//      it for five right-hand sides: the chunk's part of b, and the two columns that couple it to the separator
//      above and below.  The reciprocal square root on the serial chain is v_rsq_f64 + two Newton steps instead of
//      an IEEE square root and division (chain of ~9 instead of ~25 dependent operations per row).
//   2. one lane per dimension assembles the Schur complement on the 2 (P-1) separator unknowns from the chunks'
//      first and last two solution rows (it is SPD, block tridiagonal with 2 x 2 blocks = bandwidth 3) and solves it.
//   3. all threads: x = g - Y_top x_sep_above - Y_bottom x_sep_below, written straight to the (strided) output.
// Same arithmetic as a Cholesky solve in another elimination order: differences to the serial CPU order are rounding
// (tests: <= 1e-10 relative).  Rows stream from and to global memory as 32-byte records (two 16-byte accesses per
// lane) through register double buffers.
// Two launches: k_mlpg_chunks does step 1 with ONE wavefront per workgroup (ML_CHUNK_WGS of them at most), so the
// chunks spread over that many CUs -- as a single workgroup the sweeps were bound by one CU's memory concurrency
// (8.9 MB written by other XCDs just before, every line over the fabric: ~30 B/clk, 0.13 ms of a 0.18 ms kernel);
// k_mlpg_finish does steps 2 and 3 on a small grid, every workgroup solving the (tiny) separator system for itself
// and recovering its share of the rows.
#define ML_CHUNK_WGS 16
#define ML_MAX_CHUNKS 64      // bounds the serial separator solve (2 (P-1) rows of bandwidth 3 per dimension)
#define ML_FIN_NT 256
#define ML_FIN_WGS 8
#define ML_U 8    // rows per register buffer, forward sweep (4 values per row)
#define ML_UB 4   // backward sweep (6 values per row, 10 running values)
#define ML_G 16   // the unguarded middle section is a multiple of this many rows (2 ML_U, 4 ML_UB)
#define ML_BD 24   // doubles kept per (chunk, dimension) for step 2 (global scratch bnd[P][d][ML_BD])

__device__ __forceinline__ double ml_rsqrt(double v) {
  double r = __builtin_amdgcn_rsq(v);
  double e = fma(-v * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-v * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  return r;
}

__device__ __forceinline__ int ml_chunk_rows(const ml_part &q, int j) { return q.base + (j < q.extra ? 1 : 0); }
__device__ __forceinline__ int64_t ml_chunk_start(const ml_part &q, int j) {
  return (int64_t)j * q.base + (j < q.extra ? j : q.extra) + 2 * j;
}

__global__ __launch_bounds__(64) void k_mlpg_chunks(double *__restrict__ rec, double *__restrict__ zz,
                                                    double *__restrict__ Y, double *__restrict__ bnd, ml_dims dm,
                                                    ml_batch bt, int *__restrict__ status, long long *__restrict__ dbg) {
  // utterance blockIdx.y: its rows start at start[u] in the batch's arrays; its boundary records have their own block
  const ml_part pt = bt.u[blockIdx.y].pt;
  {
    const int64_t r0 = bt.start[blockIdx.y] * dm.d;
    rec += r0 * 4; zz += r0 * 2; Y += r0 * 4;
    bnd += (size_t)blockIdx.y * ML_CHUNK_WGS * 64 * ML_BD;
  }
  const int lane = threadIdx.x, d = dm.d, P = pt.P;
#define ML_STAMP(n) do { if (dbg && lane == 0 && blockIdx.x == 0) dbg[n] = clock64(); } while (0)
  ML_STAMP(0);
  const int cpw = 64 / d;
  const int slot = lane / d, c = lane - slot * d;
  const int j = blockIdx.x * cpw + slot;
  if (slot < cpw && j < P) {
    const int nj = ml_chunk_rows(pt, j);
    const int64_t st = ml_chunk_start(pt, j);
    const bool has_top = j > 0, has_bot = j < P - 1;
    // row i of this lane: records rq[i * 2 d] = {1/diag | P0, L1 | P1}, rq[i * 2 d + 1] = {L2 | P2, z or y of b | b};
    // zq[i * d] = z of the two top columns; yq[i * 2 d], yq[i * 2 d + 1] = y of the top and the bottom columns
    double2 *rq = (double2 *)rec + (st * d + c) * 2;
    double2 *zq = (double2 *)zz + st * d + c;
    double2 *yq = (double2 *)Y + (st * d + c) * 2;
    const int s2 = d * 2;
    // ---------------- forward: L L' = A_jj, z = L^-1 [b, top columns]; stored per row: 1/L[t][t], L[t][t-1], L[t][t-2]
    // Rows 0, 1 (which carry the top columns' right-hand sides) and the last rows (at least two: the bottom columns
    // start there) go through a guarded path; the rows between -- the same count, a multiple of ML_G, in every lane --
    // run without any divergent branch, two register buffers taking turns: while one is consumed the other one's
    // loads are in flight.  (With guards in that loop the compiler's wait counts degrade to "everything issued".)
    const int nmain = pt.base > 4 ? ((pt.base - 4) / ML_G) * ML_G : 0;      // rows [2, 2 + nmain)
    double r1 = 0, r2 = 0, l11 = 0;
    double zg1 = 0, zg2 = 0, za1 = 0, za2 = 0, zb1 = 0, zb2 = 0;
    double p2f = 0, p1f = 0, p2f1 = 0;
    bool bad = false;
    auto fwd_row = [&](int i, double p0, double p1, double p2, double b, double ra, double rb) {
      const double L2 = p2 * r2;
      const double L1 = (p1 - L2 * l11) * r1;
      double v = p0 - L2 * L2;
      v -= L1 * L1;
      if (!(v > 0.0)) { bad = true; v = 1.0; }
      const double r0 = ml_rsqrt(v);
      const double zg = ((b - L1 * zg1) - L2 * zg2) * r0;
      const double za = ((ra - L1 * za1) - L2 * za2) * r0;
      const double zb = ((rb - L1 * zb1) - L2 * zb2) * r0;
      rq[i * s2] = make_double2(r0, L1);
      rq[i * s2 + 1] = make_double2(L2, zg);
      zq[i * d] = make_double2(za, zb);
      r2 = r1; r1 = r0; l11 = L1;
      zg2 = zg1; zg1 = zg; za2 = za1; za1 = za; zb2 = zb1; zb1 = zb;
    };
    auto fwd_guarded = [&](int lo, int hi) {      // rows [lo, hi), a few at a time, not pipelined
      for (int i0 = lo; i0 < hi; i0 += ML_U) {
        double2 A[ML_U], B[ML_U];
#pragma unroll
        for (int u = 0; u < ML_U; ++u) {
          const int i = i0 + u < hi ? i0 + u : hi - 1;
          A[u] = rq[i * s2]; B[u] = rq[i * s2 + 1];
        }
#pragma unroll
        for (int u = 0; u < ML_U; ++u) {
          const int i = i0 + u;
          if (i < hi) {
            double ra = 0.0, rb = 0.0;       // the columns of the separator above: x[u], x[u+1]
            if (i == 0) { p2f = B[u].x; p1f = A[u].y; if (has_top) { ra = B[u].x; rb = A[u].y; } }
            if (i == 1) { p2f1 = B[u].x; if (has_top) rb = B[u].x; }
            fwd_row(i, A[u].x, A[u].y, B[u].x, B[u].y, ra, rb);
          }
        }
      }
    };
    auto fwd_load = [&](double2 (&A)[ML_U], double2 (&B)[ML_U], int i0, int last) {   // rows clamped to <= last
#pragma unroll
      for (int u = 0; u < ML_U; ++u) {
        const int i = i0 + u < last ? i0 + u : last;
        A[u] = rq[i * s2]; B[u] = rq[i * s2 + 1];
      }
    };
    auto fwd_rows = [&](const double2 (&A)[ML_U], const double2 (&B)[ML_U], int i0) {
#pragma unroll
      for (int u = 0; u < ML_U; ++u) fwd_row(i0 + u, A[u].x, A[u].y, B[u].x, B[u].y, 0.0, 0.0);
    };
    fwd_guarded(0, nj < 2 ? nj : 2);
    if (nmain > 0) {
      double2 A0[ML_U], B0[ML_U], A1[ML_U], B1[ML_U];
      const int last = 1 + nmain;
      fwd_load(A0, B0, 2, last);
      for (int i0 = 2; i0 < 2 + nmain; i0 += 2 * ML_U) {
        fwd_load(A1, B1, i0 + ML_U, last);
        fwd_rows(A0, B0, i0);
        fwd_load(A0, B0, i0 + 2 * ML_U, last);
        fwd_rows(A1, B1, i0 + ML_U);
      }
    }
    fwd_guarded(2 + nmain, nj);
    if (bad) atomicExch(status, 2);
    ML_STAMP(1);
    // the columns of the separator below are zero until the last two rows of the chunk
    double zc0 = 0, zc1 = 0, zd1 = 0;     // z of column x[s] at rows n-2, n-1; of column x[s+1] at row n-1
    if (has_bot) {
      const double *sq = rec + ((st + nj) * d + c) * 4;
      const double p1s = sq[1], p2s = sq[2], p2s1 = sq[4 * d + 2];
      zc0 = p2s * r2;
      zc1 = (p1s - l11 * zc0) * r1;
      zd1 = p2s1 * r1;
    }
    // ---------------- backward: y = L^-T z, five columns; the same three sections in reverse
    double *bo = bnd + ((size_t)j * d + c) * ML_BD;
    bo[20] = p2f; bo[21] = p1f; bo[22] = p2f1;
    double n11 = 0, n22 = 0, n12 = 0;
    double yg1 = 0, yg2 = 0, ya1 = 0, ya2 = 0, yb1 = 0, yb2 = 0, yc1 = 0, yc2 = 0, yd1 = 0, yd2 = 0;
    double yg, ya, yb, yc, yd;
    auto bwd_row = [&](int i, double2 A, double2 B, double2 Z, double zc, double zd) {
      const double r0 = A.x;
      yg = ((B.y - n11 * yg1) - n22 * yg2) * r0;
      ya = ((Z.x - n11 * ya1) - n22 * ya2) * r0;
      yb = ((Z.y - n11 * yb1) - n22 * yb2) * r0;
      yc = ((zc - n11 * yc1) - n22 * yc2) * r0;
      yd = ((zd - n11 * yd1) - n22 * yd2) * r0;
      ((double *)(rq + i * s2 + 1))[1] = yg;
      yq[i * s2] = make_double2(ya, yb);
      yq[i * s2 + 1] = make_double2(yc, yd);
      n22 = n12; n11 = A.y; n12 = B.x;
      yg2 = yg1; yg1 = yg; ya2 = ya1; ya1 = ya; yb2 = yb1; yb1 = yb; yc2 = yc1; yc1 = yc; yd2 = yd1; yd1 = yd;
    };
    auto bwd_guarded = [&](int lo, int hi) {      // rows hi-1 down to lo
      for (int i0 = hi - 1; i0 >= lo; i0 -= ML_UB) {
        double2 A[ML_UB], B[ML_UB], Z[ML_UB];
#pragma unroll
        for (int u = 0; u < ML_UB; ++u) {
          const int i = i0 - u >= lo ? i0 - u : lo;
          A[u] = rq[i * s2]; B[u] = rq[i * s2 + 1]; Z[u] = zq[i * d];
        }
#pragma unroll
        for (int u = 0; u < ML_UB; ++u) {
          const int i = i0 - u;
          if (i >= lo) {
            bwd_row(i, A[u], B[u], Z[u], i == nj - 1 ? zc1 : (i == nj - 2 ? zc0 : 0.0), i == nj - 1 ? zd1 : 0.0);
            if (i >= nj - 2) { double *o = bo + 15 - 5 * (nj - 1 - i); o[0] = yg; o[1] = ya; o[2] = yb; o[3] = yc; o[4] = yd; }
            if (i <= 1) { double *o = bo + 5 * i; o[0] = yg; o[1] = ya; o[2] = yb; o[3] = yc; o[4] = yd; }
          }
        }
      }
    };
    auto bwd_load = [&](double2 (&A)[ML_UB], double2 (&B)[ML_UB], double2 (&Z)[ML_UB], int i0) {   // rows i0, i0-1, ..., >= 2
#pragma unroll
      for (int u = 0; u < ML_UB; ++u) {
        const int i = i0 - u >= 2 ? i0 - u : 2;
        A[u] = rq[i * s2]; B[u] = rq[i * s2 + 1]; Z[u] = zq[i * d];
      }
    };
    auto bwd_rows = [&](const double2 (&A)[ML_UB], const double2 (&B)[ML_UB], const double2 (&Z)[ML_UB], int i0) {
#pragma unroll
      for (int u = 0; u < ML_UB; ++u) bwd_row(i0 - u, A[u], B[u], Z[u], 0.0, 0.0);
    };
    bwd_guarded(2 + nmain, nj);
    if (nmain > 0) {
      double2 A0[ML_UB], B0[ML_UB], Z0[ML_UB], A1[ML_UB], B1[ML_UB], Z1[ML_UB];
      bwd_load(A0, B0, Z0, 1 + nmain);
      for (int i0 = 1 + nmain; i0 >= 2; i0 -= 2 * ML_UB) {
        bwd_load(A1, B1, Z1, i0 - ML_UB);
        bwd_rows(A0, B0, Z0, i0);
        bwd_load(A0, B0, Z0, i0 - 2 * ML_UB);
        bwd_rows(A1, B1, Z1, i0 - ML_UB);
      }
    }
    bwd_guarded(0, nj < 2 ? nj : 2);
  }
  ML_STAMP(2);
#undef ML_STAMP
}

__global__ __launch_bounds__(ML_FIN_NT) void k_mlpg_finish(const double *__restrict__ rec, const double *__restrict__ Y,
                                                          const double *__restrict__ bnd, ml_dims dm, ml_batch bt,
                                                          int *__restrict__ status, long long *__restrict__ dbg) {
  extern __shared__ double sm[];
  const ml_part pt = bt.u[blockIdx.y].pt;
  double *__restrict__ y = bt.u[blockIdx.y].y;
  const int ldy = bt.ldy;
  {
    const int64_t r0 = bt.start[blockIdx.y] * dm.d;
    rec += r0 * 4; Y += r0 * 4;
    bnd += (size_t)blockIdx.y * ML_CHUNK_WGS * 64 * ML_BD;
  }
  const int tid = threadIdx.x, d = dm.d, P = pt.P;
#define ML_STAMP(n) do { if (dbg && tid == 0 && blockIdx.x == 0) dbg[n] = clock64(); } while (0)
  ML_STAMP(3);
  const int64_t T = bt.start[blockIdx.y + 1] - bt.start[blockIdx.y];
  const int m = 2 * (P - 1);
  double *red = sm;                    // [d][m][5]: lower band (diag, 3 sub-diagonals) and right-hand side

  // ---------------- the separators: Schur complement (bandwidth 3) on 2 (P-1) unknowns per dimension
  // assembly: one thread per (separator, dimension); R[r][q] = S[r][r-q], R[r][4] = right-hand side
  for (int e = tid; e < (P - 1) * d; e += ML_FIN_NT) {
    const int k = e / d, cc = e - k * d;
    const int na = ml_chunk_rows(pt, k);
    const int64_t s = ml_chunk_start(pt, k) + na;
    const double *A = bnd + ((size_t)k * d + cc) * ML_BD, *B = bnd + ((size_t)(k + 1) * d + cc) * ML_BD;
    const double *q0 = rec + (s * d + cc) * 4, *q1 = q0 + d * 4;   // separator rows: untouched by the chunks
    const double p0s = q0[0], p1s = q0[1], p2s = q0[2], p0s1 = q1[0], p1s1 = q1[1], p2s1 = q1[2];
    // cb(col) = C_bottom' (column of the chunk above); ct(col) = C_top' (column of the chunk below);
    // at(col) = C_top' of the chunk above applied to its own columns: the coupling to the previous separator
    auto cb0 = [&](int col) { return p2s * A[10 + col] + p1s * A[15 + col]; };
    auto cb1 = [&](int col) { return p2s1 * A[15 + col]; };
    auto ct0 = [&](int col) { return B[20] * B[col]; };
    auto ct1 = [&](int col) { return B[21] * B[col] + B[22] * B[5 + col]; };
    auto at0 = [&](int col) { return A[20] * A[col]; };
    auto at1 = [&](int col) { return A[21] * A[col] + A[22] * A[5 + col]; };
    double *R0 = red + ((size_t)cc * m + 2 * k) * 5, *R1 = R0 + 5;
    R0[0] = (p0s - cb0(3)) - ct0(1);
    R0[1] = k > 0 ? -at1(3) : 0.0;
    R0[2] = k > 0 ? -at0(3) : 0.0;
    R0[3] = 0.0;
    R0[4] = (q0[3] - cb0(0)) - ct0(0);
    R1[0] = (p0s1 - cb1(4)) - ct1(2);
    R1[1] = (p1s1 - cb1(3)) - ct1(1);
    R1[2] = k > 0 ? -at1(4) : 0.0;
    R1[3] = k > 0 ? -at0(4) : 0.0;
    R1[4] = (q1[3] - cb1(0)) - ct1(0);
  }
  __syncthreads();
  // banded Cholesky and the two sweeps, one lane per dimension; the last three rows stay in registers.
  // In place: R[r][0] = 1 / L[r][r], R[r][q] = L[r][r-q], R[r][4] = solution.
  if (m > 0 && tid < d) {
    double *R = red + (size_t)tid * m * 5;
    double ri1 = 0, ri2 = 0, ri3 = 0, a1 = 0, a2 = 0, b1 = 0, z1 = 0, z2 = 0, z3 = 0;   // a: row r-1, b: row r-2
    for (int r = 0; r < m; ++r) {
      double *Rr = R + (size_t)r * 5;
      const double s0 = Rr[0], s1 = Rr[1], s2 = Rr[2], s3 = Rr[3], bb = Rr[4];
      const double l3 = s3 * ri3;
      const double l2 = (s2 - l3 * b1) * ri2;
      const double l1 = ((s1 - l3 * a2) - l2 * a1) * ri1;
      double dd = ((s0 - l3 * l3) - l2 * l2) - l1 * l1;
      if (!(dd > 0.0)) { atomicExch(status, 2); dd = 1.0; }
      const double ri = ml_rsqrt(dd);
      const double z = (((bb - l1 * z1) - l2 * z2) - l3 * z3) * ri;
      Rr[0] = ri; Rr[1] = l1; Rr[2] = l2; Rr[3] = l3; Rr[4] = z;
      ri3 = ri2; ri2 = ri1; ri1 = ri;
      b1 = a1; a1 = l1; a2 = l2;
      z3 = z2; z2 = z1; z1 = z;
    }
    double x1 = 0, x2 = 0, x3 = 0, u1 = 0, v2 = 0, w3 = 0, v1n = 0, w1n = 0, w2n = 0;
    // u1 = L[r+1][r], v2 = L[r+2][r], w3 = L[r+3][r]
    for (int r = m - 1; r >= 0; --r) {
      double *Rr = R + (size_t)r * 5;
      const double x = (((Rr[4] - u1 * x1) - v2 * x2) - w3 * x3) * Rr[0];
      Rr[4] = x;
      // row r's sub-diagonals become: L[r][r-1] -> u1 of row r-1; L[r][r-2] -> v2 of row r-2; L[r][r-3] -> w3 of row r-3
      w3 = w2n; w2n = w1n; w1n = Rr[3];
      v2 = v1n; v1n = Rr[2];
      u1 = Rr[1];
      x3 = x2; x2 = x1; x1 = x;
    }
  }
  __syncthreads();
  ML_STAMP(4);

  // ---------------- every row: x = g - Y_top x_above - Y_bottom x_below; separator rows take the reduced solution.
  // Four elements per thread and pass, all loads issued before the first use; the multipliers of a missing
  // separator are the zero columns computed above, so the index is merely clamped.
  const int big = pt.base + 3, small_ = pt.base + 2;
  const int total = (int)(T * d);
  const int share = (total + gridDim.x - 1) / gridDim.x;
  const int e_lo = blockIdx.x * share, e_hi = min(total, e_lo + share);
  if (P == 1) {
    for (int e = e_lo + tid; e < e_hi; e += ML_FIN_NT) { const int t = e / d; y[(int64_t)t * ldy + (e - t * d)] = rec[(size_t)e * 4 + 3]; }
  } else {
    for (int e0 = e_lo + tid; e0 < e_hi; e0 += 4 * ML_FIN_NT) {
      double g[4];
      double2 ya[4], yb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * ML_FIN_NT < e_hi ? e0 + q * ML_FIN_NT : e_lo;
        g[q] = rec[(size_t)e * 4 + 3];
        ya[q] = ((const double2 *)Y)[(size_t)e * 2];
        yb[q] = ((const double2 *)Y)[(size_t)e * 2 + 1];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * ML_FIN_NT;
        if (e < e_hi) {
          const int t = e / d;
          const int cc = e - t * d;
          int jj, off, nrows;
          if (t < pt.extra * big) { jj = t / big; off = t - jj * big; nrows = pt.base + 1; }
          else {
            const int t2 = t - pt.extra * big;
            jj = pt.extra + t2 / small_;
            off = t2 - (jj - pt.extra) * small_;
            nrows = pt.base;
          }
          const double *R = red + (size_t)cc * m * 5;
          const int ka = jj > 0 ? jj - 1 : 0, kb = jj < P - 1 ? jj : P - 2;
          double v = ((g[q] - ya[q].x * R[(2 * ka) * 5 + 4]) - ya[q].y * R[(2 * ka + 1) * 5 + 4]);
          v = (v - yb[q].x * R[(2 * kb) * 5 + 4]) - yb[q].y * R[(2 * kb + 1) * 5 + 4];
          if (off >= nrows) v = R[(2 * jj + (off - nrows)) * 5 + 4];
          y[(int64_t)t * ldy + cc] = v;
        }
      }
    }
  }
  ML_STAMP(5);
#undef ML_STAMP
}

// ---- host side ------------------------------------------------------------------------------------------
// T: the frames of all utterances of the call, n: their number
static size_t ml_scratch_bytes(int64_t T, int d, int M, int n = 1) {
  const int D = 3 * d;
  return kwy_pad(sizeof(double) * ml_model_stride(D) * M) + 3 * kwy_pad(sizeof(double) * T * D) +
         kwy_pad(sizeof(double) * T * M) + kwy_pad(sizeof(int) * T) + kwy_pad(sizeof(double) * T * d * 4) +
         kwy_pad(sizeof(double) * T * d * 2) + kwy_pad(sizeof(double) * T * d * 4) +
         kwy_pad(sizeof(double) * (size_t)n * ML_CHUNK_WGS * 64 * ML_BD) + kwy_pad(64);
}

// frame-tile walkers per mixture: enough workgroups for every CU (one fits per CU), not more than tiles
static int ml_logp_waves(int D) {      // wavefronts per workgroup of k_gmm_logp: eight if Z_m and 128 frames fit the LDS
  return sizeof(double) * (size_t)(ml_np(D) + 128) * (ml_kp(D) + 1) <= 150 * 1024 ? 8 : 4;
}
static size_t ml_logp_lds(int D) {
  return sizeof(double) * (size_t)(ml_np(D) + 16 * ml_logp_waves(D)) * (ml_kp(D) + 1);
}
static int ml_logp_splits(int64_t T, int M, int D) {
  const int tile = 16 * ml_logp_waves(D);
  const int64_t ntiles = (T + tile - 1) / tile;
  int64_t s = (256 + M - 1) / M;
  if (s > ntiles) s = ntiles;
  return (int)(s < 1 ? 1 : s);
}

static int ml_launch_logp(kwy_ctx *ctx, const double *X, const ml_dims &dm, const double *model, double *logp) {
  const size_t lds = ml_logp_lds(dm.D);
  const dim3 grid((unsigned)ml_logp_splits(dm.T, dm.M, dm.D), dm.M);
  if (ml_logp_waves(dm.D) == 8) {
    KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_logp<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KWY_PROF(ctx, "k_gmm_logp", hipLaunchKernelGGL(k_gmm_logp<8>, grid, dim3(512), lds, ctx->stream, X, dm, model, logp));
  } else {
    KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_logp<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KWY_PROF(ctx, "k_gmm_logp", hipLaunchKernelGGL(k_gmm_logp<4>, grid, dim3(256), lds, ctx->stream, X, dm, model, logp));
  }
  return KWY_OK;
}

// chunks of the partitioned solve of a T-row system: 64/d per wavefront, at least 16 rows each
static ml_part ml_partition(int64_t T, int d) {
  ml_part pt;
  const int cpw = 64 / d;
  pt.P = ML_CHUNK_WGS * cpw;
  if (pt.P > ML_MAX_CHUNKS) pt.P = ML_MAX_CHUNKS;
  if ((int64_t)pt.P > T / 16) pt.P = (int)(T / 16);
  if (pt.P < 1) pt.P = 1;
  pt.base = (int)((T - 2 * (pt.P - 1)) / pt.P);
  pt.extra = (int)((T - 2 * (pt.P - 1)) % pt.P);
  return pt;
}

// One conversion call over a batch of utterances: bt.n, bt.ldx / ldy and every u[k].x / y / keep_* filled in by the
// caller, Ts[k] the utterances' frames.  `prepared`: a model made by kwy_gmm_prepare_dev (then weights / means / covs
// are unused), or null.
static int mlpg_batch_core(kwy_ctx *ctx, ml_batch &bt, const int64_t *Ts, int d, int M, const double *weights,
                           const double *means, const double *covs, int diff, int **status_out,
                           const double *prepared) {
  const int D = 3 * d, n = bt.n;
  int64_t T = 0;
  int Pmax = 1;
  for (int k = 0; k < n; ++k) {
    bt.start[k] = T;
    T += Ts[k];
    bt.u[k].pt = ml_partition(Ts[k], d);
    Pmax = std::max(Pmax, bt.u[k].pt.P);
  }
  for (int k = n; k <= KWY_BATCH_MAX; ++k) bt.start[k] = T;
  for (int k = n; k < KWY_BATCH_MAX; ++k) bt.u[k] = bt.u[0];
  ml_dims dm = {d, D, M, T};
  double *model = prepared ? const_cast<double *>(prepared) : kwy_arena<double>(ctx, ml_model_stride(D) * M);
  double *X = kwy_arena<double>(ctx, (size_t)T * D);
  double *E = kwy_arena<double>(ctx, (size_t)T * D);
  double *Dv = kwy_arena<double>(ctx, (size_t)T * D);
  double *logp = kwy_arena<double>(ctx, (size_t)T * M);
  int *mix = kwy_arena<int>(ctx, T);
  double *band = kwy_arena<double>(ctx, (size_t)T * d * 4);   // records {P0, P1, P2, rhs}
  double *rhs = kwy_arena<double>(ctx, (size_t)T * d * 2);    // forward solutions of the two top columns
  double *Ysp = kwy_arena<double>(ctx, (size_t)T * d * 4);
  double *bnd = kwy_arena<double>(ctx, (size_t)n * ML_CHUNK_WGS * 64 * ML_BD);
  int *status = kwy_arena<int>(ctx, 16);
  if (!model || !X || !E || !Dv || !logp || !mix || !band || !rhs || !Ysp || !bnd || !status) {
    ctx->err = "gmm_mlpg: scratch arena too small";
    return KWY_ENOMEM;
  }
  *status_out = status;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int) * 16, ctx->stream));
  size_t lds_prep = sizeof(double) * 3 * D * D;
  size_t lds_logp = ml_logp_lds(D);
  if (lds_prep > 160 * 1024 || lds_logp > 160 * 1024) { ctx->err = "gmm_mlpg: feature dimension too large"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
  const int cpw = 64 / d;
  const size_t lds_solve = sizeof(double) * ((size_t)d * 2 * (Pmax - 1) * 5 + 8);
  if (lds_solve > 160 * 1024) { ctx->err = "gmm_mlpg: static dimension too large"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_mlpg_finish, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_solve));
  if (!prepared)
    hipLaunchKernelGGL(k_gmm_prep, dim3(M), dim3(KWY_THREADS), lds_prep, ctx->stream, weights, means, covs, D,
                       diff, model, status);
  const unsigned ge = (unsigned)((T * d + 255) / 256);
  hipLaunchKernelGGL(k_delta, dim3(ge), dim3(256), 0, ctx->stream, bt, dm, X);
  KWY_TRY(ml_launch_logp(ctx, X, dm, model, logp));
  hipLaunchKernelGGL(k_gmm_cond, dim3((unsigned)T), dim3(128), 0, ctx->stream, X, dm, model, logp, E, Dv, mix);
  hipLaunchKernelGGL(k_mlpg_build, dim3(ge), dim3(256), 0, ctx->stream, E, Dv, bt, dm, band);
  KWY_PROF(ctx, "k_mlpg_chunks", hipLaunchKernelGGL(k_mlpg_chunks, dim3((unsigned)((Pmax + cpw - 1) / cpw), n), dim3(64), 0, ctx->stream,
                                                     band, rhs, Ysp, bnd, dm, bt, status, (long long *)ctx->dbg));
  {
    int64_t Tmax = 0;
    for (int k = 0; k < n; ++k) Tmax = std::max(Tmax, Ts[k]);
    int fin = (int)((Tmax * d + 4 * ML_FIN_NT - 1) / (4 * ML_FIN_NT));
    if (fin > ML_FIN_WGS) fin = ML_FIN_WGS;
    KWY_PROF(ctx, "k_mlpg_finish", hipLaunchKernelGGL(k_mlpg_finish, dim3((unsigned)fin, n), dim3(ML_FIN_NT), lds_solve, ctx->stream,
                                                       band, Ysp, bnd, dm, bt, status, (long long *)ctx->dbg));
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// one utterance.  x / y: T x d with rows ldx / ldy doubles apart (0: packed); keep_in / keep_out: see k_delta
static int mlpg_core(kwy_ctx *ctx, const double *x, int64_t T, int d, int M, const double *weights,
                     const double *means, const double *covs, int diff, double *y, int **status_out,
                     const double *prepared = nullptr, int ldx = 0, int ldy = 0, const double *keep_in = nullptr,
                     double *keep_out = nullptr) {
  ml_batch bt;
  bt.n = 1;
  bt.ldx = ldx > 0 ? ldx : d;
  bt.ldy = ldy > 0 ? ldy : d;
  bt.u[0].x = x; bt.u[0].y = y; bt.u[0].keep_in = keep_in; bt.u[0].keep_out = keep_out;
  return mlpg_batch_core(ctx, bt, &T, d, M, weights, means, covs, diff, status_out, prepared);
}

// x, y: T x D device rows; weights / means / covs: the joint mixture over 2 D dimensions, on the device
static int soft_core(kwy_ctx *ctx, const double *x, int64_t T, int D, int M, const double *weights,
                     const double *means, const double *covs, int diff, double *y, int **status_out) {
  ml_dims dm = {D, D, M, T};
  double *model = kwy_arena<double>(ctx, ml_model_stride(D) * M);
  double *logp = kwy_arena<double>(ctx, (size_t)T * M);
  int *status = kwy_arena<int>(ctx, 16);
  if (!model || !logp || !status) { ctx->err = "gmm_convert_frames: scratch arena too small"; return KWY_ENOMEM; }
  *status_out = status;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int) * 16, ctx->stream));
  const size_t lds_prep = sizeof(double) * 3 * D * D;
  const size_t lds_logp = ml_logp_lds(D);
  if (lds_prep > 160 * 1024 || lds_logp > 160 * 1024 || D > 192 || M > 256) {
    // (3 D^2 doubles of LDS for the preparation: 160 KB hold D <= 82)
    ctx->err = "gmm_convert_frames: feature dimension (<= 82 per side) or mixture count (<= 256) too large";
    return KWY_EINVAL;
  }
  KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
  hipLaunchKernelGGL(k_gmm_prep, dim3(M), dim3(KWY_THREADS), lds_prep, ctx->stream, weights, means, covs, D, diff, model,
                     status);
  KWY_TRY(ml_launch_logp(ctx, x, dm, model, logp));
  hipLaunchKernelGGL(k_gmm_soft, dim3((unsigned)T), dim3(128), 0, ctx->stream, x, dm, model, logp, y);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static size_t soft_scratch_bytes(int64_t T, int D, int M) {
  return kwy_pad(sizeof(double) * ml_model_stride(D) * M) + kwy_pad(sizeof(double) * T * M) + kwy_pad(64);
}

static int soft_check(kwy_ctx *ctx, const void *x, int64_t T, int D, int M, const void *w, const void *mu,
                      const void *cv, const void *y) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !w || !mu || !cv || !y || T <= 0 || D <= 0 || M <= 0) {
    ctx->err = "gmm_convert_frames: bad argument";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_gmm_convert_frames_dev(kwy_ctx *ctx, const double *x, int64_t T, int D, int M,
                                          const double *weights, const double *means, const double *covs, int diff,
                                          double *y) {
  KWY_TRY(soft_check(ctx, x, T, D, M, weights, means, covs, y));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, soft_scratch_bytes(T, D, M)));
  int *status;
  return soft_core(ctx, x, T, D, M, weights, means, covs, diff, y, &status);
}

extern "C" int kwy_gmm_convert_frames(kwy_ctx *ctx, const double *x, int64_t T, int D, int M, const double *weights,
                                      const double *means, const double *covs, int diff, double *y) {
  KWY_TRY(soft_check(ctx, x, T, D, M, weights, means, covs, y));
  KWY_HIP(hipSetDevice(ctx->device));
  const int D2 = 2 * D;
  const size_t bx = kwy_pad(sizeof(double) * T * D), bw = kwy_pad(sizeof(double) * M);
  const size_t bm = kwy_pad(sizeof(double) * M * D2), bc = kwy_pad(sizeof(double) * (size_t)M * D2 * D2);
  KWY_TRY(kwy_arena_begin(ctx, soft_scratch_bytes(T, D, M) + 2 * bx + bw + bm + bc));
  double *dx = kwy_arena<double>(ctx, (size_t)T * D), *dy = kwy_arena<double>(ctx, (size_t)T * D);
  double *dw = kwy_arena<double>(ctx, M), *dmu = kwy_arena<double>(ctx, (size_t)M * D2);
  double *dcv = kwy_arena<double>(ctx, (size_t)M * D2 * D2);
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * T * D, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dw, weights, sizeof(double) * M, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dmu, means, sizeof(double) * M * D2, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dcv, covs, sizeof(double) * (size_t)M * D2 * D2, hipMemcpyHostToDevice, ctx->stream));
  int *status;
  KWY_TRY(soft_core(ctx, dx, T, D, M, dw, dmu, dcv, diff, dy, &status));
  int hstatus = 0;
  KWY_HIP(hipMemcpyAsync(y, dy, sizeof(double) * T * D, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  if (hstatus != 0) { ctx->err = "gmm_convert_frames: source covariance is not positive definite"; return KWY_ENUMERIC; }
  return KWY_OK;
}

static int ml_check(kwy_ctx *ctx, const void *x, int64_t T, int d, int M, const void *w, const void *mu,
                    const void *cv, const void *y) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !w || !mu || !cv || !y || T <= 0 || d <= 0 || d > 64 || M <= 0) {
    ctx->err = "gmm_mlpg: bad argument";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_gmm_mlpg_dev(kwy_ctx *ctx, const double *x, int64_t T, int d, int M,
                                const double *weights, const double *means, const double *covs, int diff,
                                double *y) {
  KWY_TRY(ml_check(ctx, x, T, d, M, weights, means, covs, y));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, ml_scratch_bytes(T, d, M)));
  int *status;
  return mlpg_core(ctx, x, T, d, M, weights, means, covs, diff, y, &status);
}

extern "C" int64_t kwy_gmm_model_doubles(int d, int M) {
  if (d <= 0 || d > 64 || M <= 0) return 0;
  return (int64_t)ml_model_stride(3 * d) * M;
}

extern "C" int kwy_gmm_prepare_dev(kwy_ctx *ctx, const double *weights, const double *means, const double *covs,
                                   int d, int M, int diff, double *model) {
  if (!ctx) return KWY_EINVAL;
  if (!weights || !means || !covs || !model || d <= 0 || d > 64 || M <= 0) {
    ctx->err = "gmm_prepare: bad argument";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int D = 3 * d;
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(64)));
  int *status = kwy_arena<int>(ctx, 16);
  if (!status) { ctx->err = "gmm_prepare: scratch arena too small"; return KWY_ENOMEM; }
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int) * 16, ctx->stream));
  size_t lds_prep = sizeof(double) * 3 * D * D;
  if (lds_prep > 160 * 1024) { ctx->err = "gmm_prepare: feature dimension too large"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
  KWY_PROF(ctx, "k_gmm_prep", hipLaunchKernelGGL(k_gmm_prep, dim3(M), dim3(KWY_THREADS), lds_prep, ctx->stream, weights, means, covs, D,
                     diff, model, status));
  KWY_HIP(hipGetLastError());
  int hstatus = 0;
  KWY_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  if (hstatus != 0) { ctx->err = "gmm_prepare: source covariance is not positive definite"; return KWY_ENUMERIC; }
  return KWY_OK;
}

extern "C" int kwy_gmm_mlpg_model_dev(kwy_ctx *ctx, const double *x, int64_t T, int d, int M, const double *model,
                                      double *y) {
  KWY_TRY(ml_check(ctx, x, T, d, M, model, model, model, y));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, ml_scratch_bytes(T, d, M)));
  int *status;
  return mlpg_core(ctx, x, T, d, M, nullptr, nullptr, nullptr, 0, y, &status, model);
}

// MelCepstrumFeatureConverter.convert at the converter's own sampling rate (kwiiyatta/converter/mcep.py:47-61):
// mc, mc_out: T x (d + 1) mel-cepstra; column 0 (power) is copied, columns 1..d go through delta + GMM + MLPG.
extern "C" int kwy_convert_mcep_dev(kwy_ctx *ctx, const double *mc, int64_t T, int d, int M, const double *model,
                                    double *mc_out) {
  KWY_TRY(ml_check(ctx, mc, T, d, M, model, model, model, mc_out));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, ml_scratch_bytes(T, d, M)));
  int *status;
  return mlpg_core(ctx, mc + 1, T, d, M, nullptr, nullptr, nullptr, 0, mc_out + 1, &status, model, d + 1, d + 1, mc,
                   mc_out);
}

extern "C" int kwy_convert_mcep_batch_dev(kwy_ctx *ctx, const kwy_convert_job *jobs, int count, int d, int M,
                                          const double *model) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0) { ctx->err = "convert_mcep_batch: bad argument"; return KWY_EINVAL; }
  if (count == 0) return KWY_OK;
  size_t bytes = 0;
  for (int j0 = 0; j0 < count; j0 += KWY_BATCH_MAX) {
    int64_t T = 0;
    const int n = std::min(KWY_BATCH_MAX, count - j0);
    for (int j = j0; j < j0 + n; ++j) {
      KWY_TRY(ml_check(ctx, jobs[j].mc, jobs[j].T, d, M, model, model, model, jobs[j].mc_out));
      T += jobs[j].T;
    }
    bytes += ml_scratch_bytes(T, d, M, n);
  }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  for (int j0 = 0; j0 < count; j0 += KWY_BATCH_MAX) {
    ml_batch bt;
    int64_t Ts[KWY_BATCH_MAX];
    bt.n = std::min(KWY_BATCH_MAX, count - j0);
    bt.ldx = bt.ldy = d + 1;
    for (int k = 0; k < bt.n; ++k) {
      const kwy_convert_job &q = jobs[j0 + k];
      bt.u[k].x = q.mc + 1; bt.u[k].y = q.mc_out + 1; bt.u[k].keep_in = q.mc; bt.u[k].keep_out = q.mc_out;
      Ts[k] = q.T;
    }
    int *status;
    KWY_TRY(mlpg_batch_core(ctx, bt, Ts, d, M, nullptr, nullptr, nullptr, 0, &status, model));
  }
  return KWY_OK;
}

extern "C" int kwy_gmm_mlpg(kwy_ctx *ctx, const double *x, int64_t T, int d, int M, const double *weights,
                            const double *means, const double *covs, int diff, double *y) {
  KWY_TRY(ml_check(ctx, x, T, d, M, weights, means, covs, y));
  KWY_HIP(hipSetDevice(ctx->device));
  const int D2 = 6 * d;
  size_t bx = kwy_pad(sizeof(double) * T * d), bw = kwy_pad(sizeof(double) * M);
  size_t bm = kwy_pad(sizeof(double) * M * D2), bc = kwy_pad(sizeof(double) * (size_t)M * D2 * D2);
  KWY_TRY(kwy_arena_begin(ctx, ml_scratch_bytes(T, d, M) + 2 * bx + bw + bm + bc));
  double *dx = kwy_arena<double>(ctx, (size_t)T * d), *dy = kwy_arena<double>(ctx, (size_t)T * d);
  double *dw = kwy_arena<double>(ctx, M), *dmu = kwy_arena<double>(ctx, (size_t)M * D2);
  double *dcv = kwy_arena<double>(ctx, (size_t)M * D2 * D2);
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * T * d, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dw, weights, sizeof(double) * M, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dmu, means, sizeof(double) * M * D2, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dcv, covs, sizeof(double) * (size_t)M * D2 * D2, hipMemcpyHostToDevice, ctx->stream));
  int *status;
  KWY_TRY(mlpg_core(ctx, dx, T, d, M, dw, dmu, dcv, diff, dy, &status));
  int hstatus = 0;
  KWY_HIP(hipMemcpyAsync(y, dy, sizeof(double) * T * d, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  if (hstatus != 0) {
    ctx->err = hstatus == 1 ? "gmm_mlpg: source covariance is not positive definite"
                            : "gmm_mlpg: MLPG system is not positive definite";
    return KWY_ENUMERIC;
  }
  return KWY_OK;
}
