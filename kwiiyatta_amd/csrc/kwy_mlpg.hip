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
//   k_mlpg_solve per static dim: banded Cholesky + two triangular sweeps
//
// Algorithmic HBM bytes per frame: d*8 in, d*8 out (+ the GMM once per call).
#include <math.h>

#include "kwy_internal.hpp"

#define ML_TILE 64  // frames per k_gmm_logp workgroup

struct ml_dims { int d, D, M; int64_t T; };

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
__global__ void k_delta(const double *__restrict__ x, int ldx, ml_dims dm, double *__restrict__ X,
                        const double *__restrict__ keep_in, double *__restrict__ keep_out, int ldy) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dm.T * dm.d) return;
  const int64_t t = e / dm.d;
  const int c = (int)(e % dm.d);
  if (keep_out && c == 0) keep_out[t * ldy] = keep_in[t * ldx];
  const double xm = t > 0 ? x[(t - 1) * ldx + c] : 0.0;
  const double x0 = x[t * ldx + c];
  const double xp = t + 1 < dm.T ? x[(t + 1) * ldx + c] : 0.0;
  double *o = X + t * dm.D;
  o[c] = 0.0 + x0 * 1.0;
  double s = 0.0;
  if (t > 0) s += xm * -0.5;
  s += x0 * 0.0;
  if (t + 1 < dm.T) s += xp * 0.5;
  o[dm.d + c] = s;
  s = 0.0;
  if (t > 0) s += xm * 1.0;
  s += x0 * -2.0;
  if (t + 1 < dm.T) s += xp * 1.0;
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

__global__ __launch_bounds__(KWY_THREADS) void k_gmm_logp(const double *__restrict__ X, ml_dims dm,
                                                         const double *__restrict__ model,
                                                         double *__restrict__ logp) {
  extern __shared__ double smem[];
  const int D = dm.D, Kp = ml_kp(D), NP = ml_np(D), ZS = Kp + 1;
  double *Zs = smem;             // NP x ZS
  double *dts = Zs + NP * ZS;    // ML_TILE x ZS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.y;
  const double *mod = model + (size_t)m * ml_model_stride(D);
  const double *mZ = mod, *mux = mod + 2 * D * D;
  const double cst = mod[2 * D * D + 3 * D];
  for (int jb = 4 * wv; jb < NP; jb += 4 * KWY_WAVES) {  // four rows per wavefront and step, loads batched
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
  const int64_t ntiles = (dm.T + ML_TILE - 1) / ML_TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t t0 = tile * ML_TILE;
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

// ---- MLPG: normal equations ---------------------------------------------------------------------
// band[t][c][0..2] = P[t][t], P[t][t-1], P[t][t-2];  rhs[t][c]
__device__ __forceinline__ double ml_wcoef(int w, int k) {  // window w at offset k in [-1, 1]
  if (w == 0) return k == 0 ? 1.0 : 0.0;
  if (w == 1) return k == -1 ? -0.5 : (k == 0 ? 0.0 : 0.5);
  return k == 0 ? -2.0 : 1.0;
}

__global__ void k_mlpg_build(const double *__restrict__ E, const double *__restrict__ Dv, ml_dims dm,
                             double *__restrict__ band, double *__restrict__ rhs) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dm.T * dm.d) return;
  const int64_t a = e / dm.d;
  const int c = (int)(e % dm.d);
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, b = 0.0;
  for (int w = 0; w < 3; ++w) {
    const int l = w == 0 ? 0 : 1;
    for (int64_t t = a - l; t <= a + l; ++t) {  // frames whose window touches row a (ascending t)
      if (t < 0 || t >= dm.T) continue;
      const int k1 = (int)(a - t);
      const double prec = 1 / Dv[t * dm.D + w * dm.d + c];
      const double bm = prec * E[t * dm.D + w * dm.d + c];
      const double ca = ml_wcoef(w, k1);
      b += ca * bm;
      for (int k2 = -l; k2 <= k1; ++k2) {
        const int64_t cc = t + k2;
        if (cc < 0 || cc >= dm.T) continue;
        const double v = ca * prec * ml_wcoef(w, k2);
        const int off = (int)(a - cc);
        if (off == 0) p0 += v; else if (off == 1) p1 += v; else p2 += v;
      }
    }
  }
  band[(a * dm.d + c) * 3 + 0] = p0;
  band[(a * dm.d + c) * 3 + 1] = p1;
  band[(a * dm.d + c) * 3 + 2] = p2;
  rhs[a * dm.d + c] = b;
}

// ---- MLPG: banded Cholesky + forward/backward sweeps, one lane per static dimension ------------------
// The recurrence is serial in t and only d (24) systems wide: wavefront 0 walks the frames, one
// lane per dimension, on a tile of frames held in LDS, while wavefronts 1..3 stream the band
// through two LDS buffers around it -- write the finished tile back, fetch the next one -- so the
// walk never waits for HBM.  (One wavefront doing both spent 90 % of its time on the copies.)
#define ML_SOLVE_NT 256
__device__ __forceinline__ void ml_tile_copy(double *__restrict__ dst, const double *__restrict__ src, int n,
                                             int first, int nthreads) {
#pragma unroll 8
  for (int e = first; e < n; e += nthreads) dst[e] = src[e];
}

// y rows are ldy doubles apart
__device__ __forceinline__ void ml_rows_out(double *__restrict__ y, int ldy, const double *__restrict__ src, int rows,
                                            int d, int first, int nthreads) {
  if (ldy == d) { ml_tile_copy(y, src, rows * d, first, nthreads); return; }
  for (int e = first; e < rows * d; e += nthreads) {
    const int r = e / d, c = e - r * d;
    y[(int64_t)r * ldy + c] = src[e];
  }
}

__global__ __launch_bounds__(ML_SOLVE_NT) void k_mlpg_solve(double *__restrict__ band, double *__restrict__ rhs,
                                                           ml_dims dm, int tile, double *__restrict__ y, int ldy,
                                                           int *__restrict__ status) {
  extern __shared__ double sm[];  // 2 x { band tile [tile][d][3], rhs tile [tile][d] }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, d = dm.d;
  const int64_t T = dm.T;
  const int c = lane;
  const int bufsz = tile * d * 4;
  const int64_t ntiles = (T + tile - 1) / tile;
  auto SB = [&](int64_t k) { return sm + (k & 1) * bufsz; };
  auto SR = [&](int64_t k) { return sm + (k & 1) * bufsz + tile * d * 3; };
  auto NT_OF = [&](int64_t k) { return (int)min((int64_t)tile, T - k * tile); };

  // ---------------- forward: L L' = A, z = L^-1 b ----------------
  // stored per row: 1/L[t][t], L[t][t-1], L[t][t-2]
  double r1 = 0, l1_1 = 0, z1 = 0;  // row t-1: 1/diag, sub1; z[t-1]
  double r2 = 0, z2 = 0;            // row t-2: 1/diag; z[t-2]
  ml_tile_copy(SB(0), band, NT_OF(0) * d * 3, tid, ML_SOLVE_NT);
  ml_tile_copy(SR(0), rhs, NT_OF(0) * d, tid, ML_SOLVE_NT);
  __syncthreads();
  for (int64_t k = 0; k < ntiles; ++k) {
    const int nt = NT_OF(k);
    const int64_t t0 = k * tile;
    if (wv > 0) {
      const int h = tid - 64;
      if (k > 0) {  // tile k-1 is finished: back to memory, its buffer is then free for tile k+1
        ml_tile_copy(band + (k - 1) * tile * d * 3, SB(k - 1), tile * d * 3, h, ML_SOLVE_NT - 64);
        ml_tile_copy(rhs + (k - 1) * tile * d, SR(k - 1), tile * d, h, ML_SOLVE_NT - 64);
      }
      if (k + 1 < ntiles) {
        ml_tile_copy(SB(k + 1), band + (k + 1) * tile * d * 3, NT_OF(k + 1) * d * 3, h, ML_SOLVE_NT - 64);
        ml_tile_copy(SR(k + 1), rhs + (k + 1) * tile * d, NT_OF(k + 1) * d, h, ML_SOLVE_NT - 64);
      }
    } else if (c < d) {
      double *sb = SB(k), *sr = SR(k);
      // Row t needs 1/L[t-1][t-1] and 1/L[t-2][t-2] twice and 1/L[t][t] once: the reciprocal of
      // the diagonal is formed once per row (and stored instead of the diagonal for the backward
      // sweep), which leaves one division and one square root on the serial chain instead of three
      // divisions and a square root.  Each quotient then rounds twice instead of once
      // (~1e-16 relative per row against the CPU's divisions).
#pragma unroll 4
      for (int tt = 0; tt < nt; ++tt) {
        const int64_t t = t0 + tt;
        double *q = sb + (tt * d + c) * 3;
        double p0 = q[0], p1 = q[1], p2 = q[2];
        double L2 = 0.0, L1 = 0.0;
        if (t >= 2) L2 = p2 * r2;
        if (t >= 1) {
          double v = p1;
          if (t >= 2) v -= L2 * l1_1;  // L[t][t-2] * L[t-1][t-2]
          L1 = v * r1;
        }
        double v = p0;
        if (t >= 2) v -= L2 * L2;
        if (t >= 1) v -= L1 * L1;
        if (!(v > 0.0)) { atomicExch(status, 2); v = 1.0; }
        const double r0 = 1.0 / sqrt(v);
        double zz = sr[tt * d + c];
        if (t >= 1) zz -= L1 * z1;
        if (t >= 2) zz -= L2 * z2;
        zz = zz * r0;
        q[0] = r0; q[1] = L1; q[2] = L2;
        sr[tt * d + c] = zz;
        r2 = r1; z2 = z1;
        r1 = r0; l1_1 = L1; z1 = zz;
      }
    }
    __syncthreads();
  }
  {  // the last tile
    const int64_t k = ntiles - 1;
    ml_tile_copy(band + k * tile * d * 3, SB(k), NT_OF(k) * d * 3, tid, ML_SOLVE_NT);
    ml_tile_copy(rhs + k * tile * d, SR(k), NT_OF(k) * d, tid, ML_SOLVE_NT);
  }
  __threadfence();  // the tiles are read back below: past this CU's L1
  __syncthreads();

  // ---------------- backward: y = L^-T z ----------------
  double y1 = 0, y2 = 0, n1_1 = 0, n2_2 = 0, n1_2 = 0;  // y[t+1], y[t+2]; L[t+1][1], L[t+2][2]
  // the last tile is still in its buffer
  for (int64_t k = ntiles - 1; k >= 0; --k) {
    const int nt = NT_OF(k);
    const int64_t t0 = k * tile;
    if (wv > 0) {
      const int h = tid - 64;
      if (k + 1 < ntiles) ml_rows_out(y + (k + 1) * tile * ldy, ldy, SR(k + 1), NT_OF(k + 1), d, h, ML_SOLVE_NT - 64);
      if (k > 0) {
        ml_tile_copy(SB(k - 1), band + (k - 1) * tile * d * 3, tile * d * 3, h, ML_SOLVE_NT - 64);
        ml_tile_copy(SR(k - 1), rhs + (k - 1) * tile * d, tile * d, h, ML_SOLVE_NT - 64);
      }
    } else if (c < d) {
      const double *sb = SB(k);
      double *sr = SR(k);
#pragma unroll 4
      for (int tt = nt - 1; tt >= 0; --tt) {
        const int64_t t = t0 + tt;
        const double *q = sb + (tt * d + c) * 3;
        double v = sr[tt * d + c];
        if (t + 1 < T) v -= n1_1 * y1;
        if (t + 2 < T) v -= n2_2 * y2;
        v = v * q[0];  // q[0] holds 1/L[t][t]
        n2_2 = n1_2;  // L[t+1][2] becomes L[(t-1)+2][2]
        y2 = y1;
        n1_1 = q[1]; n1_2 = q[2];
        y1 = v;
        sr[tt * d + c] = v;
      }
    }
    __syncthreads();
  }
  ml_rows_out(y, ldy, SR(0), NT_OF(0), d, tid, ML_SOLVE_NT);
}

// ---- host side ------------------------------------------------------------------------------------------
static size_t ml_scratch_bytes(int64_t T, int d, int M) {
  const int D = 3 * d;
  return kwy_pad(sizeof(double) * ml_model_stride(D) * M) + 3 * kwy_pad(sizeof(double) * T * D) +
         kwy_pad(sizeof(double) * T * M) + kwy_pad(sizeof(int) * T) + kwy_pad(sizeof(double) * T * d * 3) +
         kwy_pad(sizeof(double) * T * d) + kwy_pad(64);
}

// frame-tile walkers per mixture: enough workgroups for every CU (one fits per CU), not more than tiles
static int ml_logp_splits(int64_t T, int M) {
  const int64_t ntiles = (T + ML_TILE - 1) / ML_TILE;
  int64_t s = (256 + M - 1) / M;
  if (s > ntiles) s = ntiles;
  return (int)(s < 1 ? 1 : s);
}

// `prepared`: a model made by kwy_gmm_prepare_dev (then weights/means/covs are unused), or null
// x / y: T x d with rows ldx / ldy doubles apart (0: packed); keep_in / keep_out: see k_delta
static int mlpg_core(kwy_ctx *ctx, const double *x, int64_t T, int d, int M, const double *weights,
                     const double *means, const double *covs, int diff, double *y, int **status_out,
                     const double *prepared = nullptr, int ldx = 0, int ldy = 0, const double *keep_in = nullptr,
                     double *keep_out = nullptr) {
  if (ldx <= 0) ldx = d;
  if (ldy <= 0) ldy = d;
  const int D = 3 * d;
  ml_dims dm = {d, D, M, T};
  double *model = prepared ? const_cast<double *>(prepared) : kwy_arena<double>(ctx, ml_model_stride(D) * M);
  double *X = kwy_arena<double>(ctx, (size_t)T * D);
  double *E = kwy_arena<double>(ctx, (size_t)T * D);
  double *Dv = kwy_arena<double>(ctx, (size_t)T * D);
  double *logp = kwy_arena<double>(ctx, (size_t)T * M);
  int *mix = kwy_arena<int>(ctx, T);
  double *band = kwy_arena<double>(ctx, (size_t)T * d * 3);
  double *rhs = kwy_arena<double>(ctx, (size_t)T * d);
  int *status = kwy_arena<int>(ctx, 16);
  if (!model || !X || !E || !Dv || !logp || !mix || !band || !rhs || !status) {
    ctx->err = "gmm_mlpg: scratch arena too small";
    return KWY_ENOMEM;
  }
  *status_out = status;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int) * 16, ctx->stream));
  size_t lds_prep = sizeof(double) * 3 * D * D;
  size_t lds_logp = sizeof(double) * (size_t)(ml_np(D) + ML_TILE) * (ml_kp(D) + 1);
  if (lds_prep > 160 * 1024 || lds_logp > 160 * 1024) { ctx->err = "gmm_mlpg: feature dimension too large"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
  KWY_HIP(hipFuncSetAttribute((const void *)k_gmm_logp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_logp));
  // frames per LDS tile of the solve: two buffers of tile x d x 4 doubles within 128 KB
  int solve_tile = 64;
  while (solve_tile > 4 && (size_t)2 * solve_tile * d * 4 * sizeof(double) > 128 * 1024) solve_tile /= 2;
  const size_t lds_solve = (size_t)2 * solve_tile * d * 4 * sizeof(double);
  if (lds_solve > 160 * 1024) { ctx->err = "gmm_mlpg: static dimension too large"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_mlpg_solve, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_solve));
  if (!prepared)
    hipLaunchKernelGGL(k_gmm_prep, dim3(M), dim3(KWY_THREADS), lds_prep, ctx->stream, weights, means, covs, D,
                       diff, model, status);
  const unsigned ge = (unsigned)((T * d + 255) / 256);
  hipLaunchKernelGGL(k_delta, dim3(ge), dim3(256), 0, ctx->stream, x, ldx, dm, X, keep_in, keep_out, ldy);
  KWY_PROF(ctx, "k_gmm_logp", hipLaunchKernelGGL(k_gmm_logp, dim3((unsigned)ml_logp_splits(T, M), M), dim3(KWY_THREADS), lds_logp,
                     ctx->stream, X, dm, model, logp));
  hipLaunchKernelGGL(k_gmm_cond, dim3((unsigned)T), dim3(128), 0, ctx->stream, X, dm, model, logp, E, Dv, mix);
  hipLaunchKernelGGL(k_mlpg_build, dim3(ge), dim3(256), 0, ctx->stream, E, Dv, dm, band, rhs);
  KWY_PROF(ctx, "k_mlpg_solve", hipLaunchKernelGGL(k_mlpg_solve, dim3(1), dim3(ML_SOLVE_NT), lds_solve, ctx->stream, band, rhs, dm, solve_tile, y, ldy, status));
  KWY_HIP(hipGetLastError());
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
