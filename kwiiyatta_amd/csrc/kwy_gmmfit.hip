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

#include <type_traits>

#include "kwy_internal.hpp"


__host__ __device__ static inline size_t fit_tri(int D) { return (size_t)D * (D + 1) / 2; }
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // j <= i

// per mixture: Zp[m] = packed lower-triangular inverse of chol(cov_m), cst[m] = log w - 0.5 D log 2pi - sum log L_ii
// One workgroup per mixture, the packed triangle in LDS, thread i owns row i (D <= 160 < 256 threads).
// Cholesky, left-looking: column j of L is a dot product per row, the pivot of column j+1 travels with
// the row that owns it (one barrier per column).  Inverse, row by row in place: row i of Z = L^-1 needs
// row i of L and the rows of Z above it, and takes the place of row i of L.
__global__ __launch_bounds__(KWY_THREADS) void k_fit_prec(const double *__restrict__ weights,
                                                         const double *__restrict__ covs, int D,
                                                         double *__restrict__ Zp, double *__restrict__ cst,
                                                         int *__restrict__ status) {
  extern __shared__ double L[];  // packed lower triangle, then pivs[D]
  const int tid = threadIdx.x, m = blockIdx.x, i = tid;
  const double *C = covs + (size_t)m * D * D;
  const int nt = (int)fit_tri(D);
  double *pivs = L + nt;
  for (int r = 0; r < D; ++r)
    for (int c = tid; c <= r; c += KWY_THREADS) L[tri(r, c)] = C[(size_t)r * D + c];
  __syncthreads();
  double ri = i < D ? L[tri(i, i)] : 0.0;   // C[i][i] - sum_k L[i][k]^2 over the columns done so far
  if (tid == 0) pivs[0] = ri;
  __syncthreads();
  for (int j = 0; j < D; ++j) {
    const double piv = pivs[j];
    if (!(piv > 0.0)) { if (tid == 0) atomicExch(status, 1); return; }   // uniform: every thread reads pivs[j]
    const double dj = sqrt(piv);
    if (i > j && i < D) {
      const double *li = L + tri(i, 0), *lj = L + tri(j, 0);
      double sacc = li[j];
#pragma unroll 8
      for (int k = 0; k < j; ++k) sacc -= li[k] * lj[k];
      const double v = sacc / dj;
      L[tri(i, j)] = v;
      ri -= v * v;
      if (i == j + 1) pivs[j + 1] = ri;
    } else if (i == j) {
      L[tri(j, j)] = dj;
    }
    __syncthreads();
  }
  for (int r = 0; r < D; ++r) {
    const double *lr = L + tri(r, 0);
    const double lrr = lr[r];
    double z = 0.0;
    if (i < r) {
      double sacc = 0.0;
#pragma unroll 8
      for (int k = i; k < r; ++k) sacc += lr[k] * L[tri(k, i)];
      z = -sacc / lrr;
    } else if (i == r) {
      z = 1.0 / lrr;
    }
    __syncthreads();   // row r of L has been read by everyone
    if (i <= r) L[tri(r, i)] = z;
    __syncthreads();
  }
  double *zp = Zp + (size_t)m * nt;
  for (int e = tid; e < nt; e += KWY_THREADS) zp[e] = L[e];
  if (tid == 0) {
    double ld = 0.0;
    for (int r = 0; r < D; ++r) ld += log(L[tri(r, r)]);   // Z_rr = 1 / L_rr
    cst[m] = -0.5 * (D * log(2.0 * KWY_PI)) + ld + log(weights[m]);
  }
}

// weighted log prob wlp[t][m] = cst[m] - 0.5 * || Z_m x_t - Z_m mu_m ||^2   (sklearn's form:
// np.dot(X, prec_chol) - np.dot(mu, prec_chol))
// The n x D by D x D (lower-triangular) product runs on v_mfma_f64_16x16x4_f64 (2048 flop per
// instruction on two 512-byte operands: fed from LDS alone it is LDS-bound, so the frame operand
// stays in registers).  A workgroup of eight wavefronts owns one mixture: Z is expanded once into
// MFMA-fragment order in LDS (one conflict-free 512-byte read per B operand, the entries above the
// diagonal stored as zeros: no masking in the loop; 92 KB for D = 144).  A wavefront takes 32 frames at
// a time: it loads their D columns straight from global memory into the A registers of two 16-frame
// sub-tiles (2 x 36 doubles per lane), and every B fragment it reads feeds two MFMAs.  Per 16-column
// block of Z only the k-steps up to the block's diagonal are issued (180 instead of 324 fragments for
// D = 144).  The second wavefront of each SIMD computes while the first one waits for its rows.
// Lane map: A[row l&15][k l>>4], B[k l>>4][col l&15], D[row (l>>4)+4r][col l&15].
typedef double fit_v4f64 __attribute__((ext_vector_type(4)));
#define FIT_LP_NT 512
#define FIT_LP_COLS 3   // feature columns per lane when staging a row (k_fit_cov): D <= 192

template <int NBLK>
__global__ __launch_bounds__(FIT_LP_NT) void k_fit_logprob(const double *__restrict__ X, int64_t n, int D, int M,
                                                          int nsplit, const double *__restrict__ means,
                                                          const double *__restrict__ Zp,
                                                          const double *__restrict__ cst,
                                                          double *__restrict__ wlp) {
  constexpr int KS = 4 * NBLK, NFRAG = 2 * NBLK * (NBLK + 1);
  extern __shared__ double sm[];
  double *zf = sm;                  // NFRAG x 64: fragment (nb, ks), ks < 4 nb + 4, at 2 nb (nb + 1) + ks
  double *cs = zf + NFRAG * 64;     // 16 NBLK: Z mu
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ar = lane & 15, ak = lane >> 4;
  int m, split;
  if ((nsplit & 7) == 0) {   // consecutive workgroups of one XCD (ids k, k+8, ...) = mixtures of one split
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    m = q % M;
    split = xcd + 8 * (q / M);
  } else {
    m = blockIdx.x % M;
    split = blockIdx.x / M;
  }
  const int nt = (int)fit_tri(D);
  const double *zp = Zp + (size_t)m * nt, *mu = means + (size_t)m * D;
  for (int idx = tid; idx < NFRAG * 64; idx += FIT_LP_NT) {
    const int f = idx >> 6, l = idx & 63;
    int nb = 0;
    while (2 * (nb + 1) * (nb + 2) <= f) ++nb;
    const int ks = f - 2 * nb * (nb + 1);
    const int j = 16 * nb + (l & 15), i = 4 * ks + (l >> 4);
    zf[idx] = (j < D && i <= j) ? zp[tri(j, i)] : 0.0;
  }
  for (int j = tid; j < 16 * NBLK; j += FIT_LP_NT) {
    double c = 0.0;
    if (j < D)
      for (int i = 0; i <= j; ++i) c += zp[tri(j, i)] * mu[i];
    cs[j] = c;
  }
  const double cm = cst[m];
  __syncthreads();
  const int64_t ntiles = (n + 255) / 256;
  const int64_t per = (ntiles + nsplit - 1) / nsplit;
  const int64_t tile0 = split * per, tile1 = min(ntiles, tile0 + per);
  for (int64_t tile = tile0; tile < tile1; ++tile) {
    const int64_t t0 = tile * 256 + 32 * wv;
    if (t0 >= n) break;
    // columns beyond D (last block only) meet zero fragments: clamping them keeps the loads in bounds
    const double *xa = X + min(t0 + ar, n - 1) * D, *xb = X + min(t0 + 16 + ar, n - 1) * D;
    double a0[KS], a1[KS];
#pragma unroll
    for (int ks = 0; ks < KS - 4; ++ks) {
      a0[ks] = xa[4 * ks + ak];
      a1[ks] = xb[4 * ks + ak];
    }
#pragma unroll
    for (int ks = KS - 4; ks < KS; ++ks) {
      const int c = min(4 * ks + ak, D - 1);
      a0[ks] = xa[c];
      a1[ks] = xb[c];
    }
    double q0[4] = {0.0, 0.0, 0.0, 0.0}, q1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      fit_v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      const double *frag = zf + (2 * nb * (nb + 1)) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 4 * nb + 4; ++ks) {
        const double b = frag[ks * 64];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b, acc1, 0, 0, 0);
      }
      const double c = cs[16 * nb + ar];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double y0 = acc0[r] - c, y1 = acc1[r] - c;
        q0[r] += y0 * y0;
        q1[r] += y1 * y1;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = q0[r], w = q1[r];
      v += kwy_dpp_f64<0x111>(v); w += kwy_dpp_f64<0x111>(w);
      v += kwy_dpp_f64<0x112>(v); w += kwy_dpp_f64<0x112>(w);
      v += kwy_dpp_f64<0x114>(v); w += kwy_dpp_f64<0x114>(w);
      v += kwy_dpp_f64<0x118>(v); w += kwy_dpp_f64<0x118>(w);
      q0[r] = v; q1[r] = w;
    }
    if (ar == 15) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t t = t0 + ak + 4 * r;
        if (t < n) wlp[t * M + m] = cm - 0.5 * q0[r];
        if (t + 16 < n) wlp[(t + 16) * M + m] = cm - 0.5 * q1[r];
      }
    }
  }
}

// per frame: log-sum-exp over the mixtures; wlp is overwritten by the responsibilities;
// per-block sums of the frame log-likelihoods go to ll_part[blockIdx.x].
// Sixteen lanes (one DPP row) share a frame, lane c of the row holds mixtures c, c+16, ...: a
// wavefront's loads cover four whole rows of wlp, and max and sum are rotations within the row.
template <class OP>
__device__ __forceinline__ double fit_row_allreduce(double v, OP op) {
  v = op(v, kwy_dpp_f64<0x121>(v));   // row_ror:1
  v = op(v, kwy_dpp_f64<0x122>(v));
  v = op(v, kwy_dpp_f64<0x124>(v));
  v = op(v, kwy_dpp_f64<0x128>(v));
  return v;
}
#define FIT_RESP_Q 16   // mixtures per lane: M <= 256

__global__ __launch_bounds__(KWY_THREADS) void k_fit_resp(double *__restrict__ wlp, int64_t n, int M,
                                                         double *__restrict__ ll_part) {
  __shared__ double red[8];
  const int c = threadIdx.x & 15, f = threadIdx.x >> 4;
  double lsum = 0.0;
  for (int pass = 0; pass < KWY_THREADS / 16; ++pass) {
    const int64_t t = (int64_t)blockIdx.x * KWY_THREADS + 16 * pass + f;
    const bool live = t < n;
    double *w = wlp + (live ? t : 0) * M;
    double v[FIT_RESP_Q];
    double mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < FIT_RESP_Q; ++q) {
      const int m = c + 16 * q;
      v[q] = (live && m < M) ? w[m] : -INFINITY;
      mx = fmax(mx, v[q]);
    }
    mx = fit_row_allreduce(mx, [](double a, double b) { return fmax(a, b); });
    double sm = 0.0;
#pragma unroll
    for (int q = 0; q < FIT_RESP_Q; ++q)
      if (16 * q < M) sm += exp(v[q] - mx);
    sm = fit_row_allreduce(sm, [](double a, double b) { return a + b; });
    const double lse = mx + log(sm);
#pragma unroll
    for (int q = 0; q < FIT_RESP_Q; ++q) {
      const int m = c + 16 * q;
      if (live && m < M) w[m] = exp(v[q] - lse);
    }
    if (live && c == 0) lsum += lse;
  }
  const double tot = kwy_block_sum(lsum, red);
  if (threadIdx.x == 0) ll_part[blockIdx.x] = tot;
}

// local statistics: part[chunk][m][0] = sum_t r, part[chunk][m][1+i] = sum_t r x_i over the chunk's rows.
// resp' X on v_mfma_f64_16x16x4_f64 with the frame index as k: A[mixture][frame] and B[frame][feature]
// are both rows of 16 consecutive doubles per frame, loaded straight from global memory FIT_SUM_PF
// steps ahead.  Wavefront w of workgroup (chunk, g) owns mixtures 16 (4 g + w) .. +15 and all NBLK
// feature blocks; the four wavefronts read the same rows of X (L1).
#define FIT_SUM_PF 4
template <int NBLK>
__global__ __launch_bounds__(KWY_THREADS) void k_fit_sums(const double *__restrict__ X,
                                                         const double *__restrict__ resp, int64_t n, int D,
                                                         int M, int rows_per_chunk, double *__restrict__ part,
                                                         const long long *__restrict__ gate,
                                                         const int *__restrict__ labels) {
  // labels (k-means): the weights are the one-hot rows of the labels, formed here -- 4 bytes per frame instead of a
  // row of M doubles written by the assignment and read back (the same products and sums: x times 1.0 or 0.0)
  if (gate && gate[0]) return;    // (the device-driven Lloyd loop has stopped: kwy_km_lloyd_dev)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ar = lane & 15, ak = lane >> 4;
  const int rb = 4 * blockIdx.y + wv;
  if (16 * rb >= M) return;       // no barrier in this kernel
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  const int mrow = min(16 * rb + ar, M - 1);
  const bool mok = 16 * rb + ar < M;
  const int clast = min(16 * (NBLK - 1) + ar, D - 1);
  fit_v4f64 acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc[b] = fit_v4f64{0.0, 0.0, 0.0, 0.0};
  double sumr = 0.0;
  const int64_t nks = (r1 - r0 + 3) / 4;
  double xs[FIT_SUM_PF][NBLK], as[FIT_SUM_PF];
  auto load = [&](double *xq, double &aq, int64_t ks) {
    const int64_t t = r0 + 4 * ks + ak;
    const int64_t tc = min(t, r1 - 1);
    const double *px = X + tc * D;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) xq[b] = px[b == NBLK - 1 ? clast : 16 * b + ar];
    if (labels) aq = (t < r1 && mok && labels[tc] == mrow) ? 1.0 : 0.0;
    else aq = (t < r1 && mok) ? resp[tc * M + mrow] : 0.0;
  };
#pragma unroll
  for (int p = 0; p < FIT_SUM_PF; ++p) load(xs[p], as[p], p);
  for (int64_t ks = 0; ks < nks; ks += FIT_SUM_PF) {
#pragma unroll
    for (int p = 0; p < FIT_SUM_PF; ++p) {
      double x[NBLK];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) x[b] = xs[p][b];
      const double a = as[p];     // zero behind the chunk's last frame
      load(xs[p], as[p], ks + p + FIT_SUM_PF);
      sumr += a;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[b], acc[b], 0, 0, 0);
    }
  }
  sumr += __shfl_xor(sumr, 16);
  sumr += __shfl_xor(sumr, 32);
  double *out = part + (size_t)blockIdx.x * M * (D + 1);
  if (ak == 0 && mok) out[(size_t)(16 * rb + ar) * (D + 1)] = sumr;
#pragma unroll
  for (int b = 0; b < NBLK; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * rb + ak + 4 * r, i = 16 * b + ar;
      if (m < M && i < D) out[(size_t)m * (D + 1) + 1 + i] = acc[b][r];
    }
}

// out[e] = sum_c part[c][e]
__global__ void k_fit_reduce(const double *__restrict__ part, int nchunks, int64_t len, double *__restrict__ out,
                             const long long *__restrict__ gate = nullptr) {
  if (gate && gate[0]) return;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(size_t)c * len + e];
  out[e] = s;
}

// local statistics: cpart[split][m] = sum over the split's rows of r (x - mu_m)(x - mu_m)'
// A D x n by n x D product with the frame index as the MFMA's k: v_mfma_f64_16x16x4_f64 consumes four
// frames per instruction, A = r_t d_t[i], B = d_t[j] -- the same lane layout, so one LDS read serves a
// block index both as row and as column operand.  A workgroup owns (mixture, row split) and walks over
// 32-frame tiles of d = x - mu, double-buffered in LDS (one barrier per tile, the next tile's rows are
// in flight in registers meanwhile).  Only the 16x16 blocks (I, J <= I) of the lower triangle are
// computed; they are dealt to the four wavefronts in row-major order (45 blocks -> 12/11/11/11 for
// D = 144), so a wavefront holds at most 12 accumulator blocks (96 registers): two workgroups per CU,
// every register an architectural VGPR.  Workgroups are numbered so that the 64 that run together on
// one XCD (2 per CU) are the mixtures of one row split and share its frames in that XCD's L2.
#define FIT_COV_TILE 32
#define FIT_R_MIN 6.223015277861142e-61      // 2^-200
#define FIT_R_REL 8.470329472543003e-22      // 2^-70
constexpr int fit_blk_row(int e) { int I = 0; while ((I + 1) * (I + 2) / 2 <= e) ++I; return I; }
constexpr int fit_blk_col(int e) { return e - fit_blk_row(e) * (fit_blk_row(e) + 1) / 2; }
constexpr int fit_blk_lo(int nblk, int w) { return (nblk * (nblk + 1) / 2) * w / 4; }
// does wavefront w of an nblk-block problem touch block index b (as row or column operand)?
constexpr bool fit_blk_uses(int nblk, int w, int b, bool as_row) {
  for (int e = fit_blk_lo(nblk, w); e < fit_blk_lo(nblk, w + 1); ++e)
    if ((as_row ? fit_blk_row(e) : fit_blk_col(e)) == b) return true;
  return false;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void fit_static_for(F &&f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    fit_static_for<I0 + 1, I1>(f);
  }
}

template <int NBLK, int W, int CMAX>
__device__ __forceinline__ void fit_cov_tile(const double *__restrict__ tile, const double *__restrict__ rt, int ZS,
                                             int ar, int ak, double r_min, fit_v4f64 (&acc)[CMAX]) {
  constexpr int E0 = fit_blk_lo(NBLK, W), E1 = fit_blk_lo(NBLK, W + 1);
#pragma unroll 2
  for (int ks = 0; ks < FIT_COV_TILE / 4; ++ks) {
    const double *drow = tile + (4 * ks + ak) * ZS + ar;
    const double rr = rt[4 * ks + ak];
    // Four frames whose responsibilities for this mixture are all below r_min add nothing the covariance can hold: the
    // step's operand loads and matrix instructions are skipped, uniformly for the wavefront.  r_min = 2^-200 always
    // (n <= 2^31 such terms against nk >= 10 eps change an entry by less than 2^-90 |x - mu|^2), and 2^-70 of the
    // mixture's own mass nk when the caller hands the sums over (the skipped terms then weigh less than n 2^-70 of
    // the mixture: with n <= 2^25 frames below the rounding of nk itself).  A fitted mixture in D = 144 is SPARSE in
    // this sense -- most frames belong to one or two of the 64 components.
    if (__ballot(rr >= r_min) == 0) continue;
    double dv[NBLK], av[NBLK];
    fit_static_for<0, NBLK>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr bool as_row = fit_blk_uses(NBLK, W, b, true), as_col = fit_blk_uses(NBLK, W, b, false);
      if constexpr (as_row || as_col) dv[b] = drow[16 * b];
      if constexpr (as_row) av[b] = rr * dv[b];
    });
    fit_static_for<E0, E1>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      constexpr int I = fit_blk_row(e), J = fit_blk_col(e);
      acc[e - E0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], dv[J], acc[e - E0], 0, 0, 0);
    });
  }
}

template <int NBLK, int W, int CMAX>
__device__ __forceinline__ void fit_cov_store(double *__restrict__ out, int D, int ar, int ak,
                                              const fit_v4f64 (&acc)[CMAX]) {
  constexpr int E0 = fit_blk_lo(NBLK, W), E1 = fit_blk_lo(NBLK, W + 1);
  fit_static_for<E0, E1>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    constexpr int I = fit_blk_row(e), J = fit_blk_col(e);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * I + ak + 4 * r, j = 16 * J + ar;
      if (i < D && j < D) out[(size_t)i * D + j] = acc[e - E0][r];
    }
  });
}

// act[m][g]: does any of the four frames 4 g .. 4 g + 3 (one matrix instruction's worth) carry a responsibility >= r_min
// for mixture m?  A workgroup takes FIT_ACT_GROUPS groups x all mixtures: resp is read once, coalesced along the
// mixtures, the flags leave through LDS so that they are written coalesced along the groups.
#define FIT_ACT_GROUPS 64
__global__ __launch_bounds__(KWY_THREADS) void k_fit_active(const double *__restrict__ resp, int64_t n, int D, int M,
                                                           const double *__restrict__ stats, int64_t ngroups,
                                                           unsigned char *__restrict__ act) {
  extern __shared__ unsigned char fl[];          // M x FIT_ACT_GROUPS
  const int tid = threadIdx.x;
  const int64_t g0 = (int64_t)blockIdx.x * FIT_ACT_GROUPS;
  for (int idx = tid; idx < M * FIT_ACT_GROUPS; idx += KWY_THREADS) {
    const int m = idx % M, q = idx / M;
    const int64_t g = g0 + q;
    bool any = false;
    if (g < ngroups) {
      const double r_min = fmax(FIT_R_MIN, stats ? stats[(size_t)m * (D + 1)] * FIT_R_REL : 0.0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t t = 4 * g + r;
        if (t < n && resp[t * M + m] >= r_min) any = true;
      }
    }
    fl[m * FIT_ACT_GROUPS + q] = any ? 1 : 0;
  }
  __syncthreads();
  for (int idx = tid; idx < M * FIT_ACT_GROUPS; idx += KWY_THREADS) {
    const int m = idx / FIT_ACT_GROUPS, q = idx % FIT_ACT_GROUPS;
    if (g0 + q < ngroups) act[(size_t)m * ngroups + g0 + q] = fl[idx];
  }
}

// The active groups of every mixture as a LIST (ascending): one workgroup per mixture counts and scans its flags and
// writes list[m][0 .. count[m]).  k_fit_cov builds its 32-frame tiles from eight LISTED groups each -- wherever they
// lie -- and gives every split of a mixture an equal run of consecutive tiles: every matrix instruction issued has at
// least one frame that counts, and the splits of a mixture have the same amount to do.  (Until round 5's last day a split owned a
// row range and visited the 32-row tiles of it that had any weight: on rows in arbitrary order 40 % of all tiles for one
// live group each, on rows grouped by cluster one split per component did all the work.)
__global__ __launch_bounds__(KWY_THREADS) void k_fit_active_list(const unsigned char *__restrict__ act, int64_t ntiles,
                                                                int32_t *__restrict__ list, int32_t *__restrict__ count) {   // (ntiles: entries per mixture = groups)
  __shared__ uint64_t tot[KWY_THREADS];
  __shared__ uint64_t total;
  const int m = blockIdx.x, t = threadIdx.x;
  const unsigned char *am = act + (size_t)m * ntiles;
  int32_t *lm = list + (size_t)m * ntiles;
  const int64_t chunk = (ntiles + KWY_THREADS - 1) / KWY_THREADS;
  const int64_t b0 = t * chunk, b1 = min(ntiles, b0 + chunk);
  uint64_t run = 0;
  for (int64_t i = b0; i < b1; ++i) run += am[i] ? 1 : 0;
  tot[t] = run;
  __syncthreads();
  if (t == 0) {
    uint64_t acc = 0;
    for (int i = 0; i < KWY_THREADS; ++i) { const uint64_t c = tot[i]; tot[i] = acc; acc += c; }
    total = acc;
  }
  __syncthreads();
  uint64_t at = tot[t];
  for (int64_t i = b0; i < b1; ++i)
    if (am[i]) lm[at++] = (int32_t)i;
  if (t == 0) count[m] = (int32_t)total;
}

template <int NBLK>
__global__ __launch_bounds__(KWY_THREADS, 2) void k_fit_cov(const double *__restrict__ X, const double *__restrict__ resp,
                                                           int64_t n, int D, int M, int nsplit,
                                                           const double *__restrict__ means,
                                                           const double *__restrict__ stats,
                                                           const int32_t *__restrict__ list,
                                                           const int32_t *__restrict__ count, int64_t ntiles,
                                                           double *__restrict__ cpart) {
  constexpr int NP = 16 * NBLK, ZS = NP + 1, CMAX = (NBLK * (NBLK + 1) / 2 + 3) / 4;
  constexpr int ROWS = FIT_COV_TILE / 4;   // rows of a tile staged by one wavefront
  extern __shared__ double sm[];
  double *ds = sm;                              // 2 x FIT_COV_TILE x ZS
  double *rs = ds + 2 * FIT_COV_TILE * ZS;      // 2 x FIT_COV_TILE
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ar = lane & 15, ak = lane >> 4;
  // workgroup -> (mixture, split): consecutive workgroups of one XCD (ids k, k+8, ...) = mixtures of one split
  int m, split;
  if ((nsplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    m = q % M;
    split = xcd + 8 * (q / M);
  } else {
    m = blockIdx.x % M;
    split = blockIdx.x / M;
  }
  const double *mu = means + (size_t)m * D;
  const double r_min = fmax(FIT_R_MIN, stats ? stats[(size_t)m * (D + 1)] * FIT_R_REL : 0.0);
  double muv[FIT_LP_COLS];
#pragma unroll
  for (int c = 0; c < FIT_LP_COLS; ++c) muv[c] = (lane + 64 * c < D) ? mu[lane + 64 * c] : 0.0;
  fit_v4f64 acc[CMAX];
#pragma unroll
  for (int a = 0; a < CMAX; ++a) acc[a] = fit_v4f64{0.0, 0.0, 0.0, 0.0};
  // this wavefront stages rows ROWS*w .. ROWS*w+ROWS-1 of every tile: fetched one tile ahead into registers
  double xv[ROWS][FIT_LP_COLS], rv = 0.0;
  // tile j of the mixture = the listed groups 8 j .. 8 j + 7 (four frames each; entries beyond the list: no frames)
  const int32_t *__restrict__ lm = list + (size_t)m * ntiles;
  const int cnt = count[m];                      // listed groups
  const int ntl = (cnt + 7) / 8;                 // tiles
  // (the list entries of a tile are read one tile AHEAD of its rows: the rows' addresses depend on them, and two
  // dependent trips to memory per tile are more than a tile's 2.4 us of matrix instructions cover)
  int ea = 0, eb = 0, er = 0;                    // first frames' groups of this wavefront's rows / of row tid
  auto entries = [&](int j) {
    const int e0 = 8 * j + 2 * wv, e2 = 8 * j + ((tid & (FIT_COV_TILE - 1)) >> 2);
    ea = e0 < cnt ? lm[e0] : -1;
    eb = e0 + 1 < cnt ? lm[e0 + 1] : -1;
    er = e2 < cnt ? lm[e2] : -1;
  };
  auto fetch = [&]() {                           // the rows of the tile whose entries are in (ea, eb, er)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int g = (r < 4) ? ea : eb;
      const int64_t t = g >= 0 ? 4 * (int64_t)g + (r & 3) : n;
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        xv[r][c] = (t < n && i < D) ? X[t * D + i] : muv[c];
      }
    }
    rv = 0.0;
    if (tid < FIT_COV_TILE) {
      const int64_t t = er >= 0 ? 4 * (int64_t)er + (tid & 3) : n;
      if (t < n) rv = resp[t * M + m];
    }
  };
  auto stage = [&](int buf) {
    double *dt = ds + buf * FIT_COV_TILE * ZS;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < FIT_LP_COLS; ++c) {
        const int i = lane + 64 * c;
        if (i < NP) dt[(ROWS * wv + r) * ZS + i] = xv[r][c] - muv[c];
      }
    if (tid < FIT_COV_TILE) rs[buf * FIT_COV_TILE + tid] = rv;
  };
  // split s takes the tiles [ntl s / nsplit, ntl (s + 1) / nsplit) of the mixture -- equal shares, and CONSECUTIVE ones:
  // where the posteriors are broad the same rows are live for many mixtures, and the workgroups of a split (one XCD)
  // then read them from that XCD's L2 as they did when a split owned a row range (dealing the tiles out round-robin
  // instead was measured: 7.7 against 5.2 ms per launch on config 5's corpus -- every workgroup fetched its own rows).
  // The next tile is fetched while the current one is multiplied.
  int j = (int)((int64_t)ntl * split / nsplit);
  const int jend = (int)((int64_t)ntl * (split + 1) / nsplit);
  if (j < jend) {
    entries(j);
    fetch();
    if (j + 1 < jend) entries(j + 1);
    stage(0);
    __syncthreads();
    int buf = 0;
    while (j < jend) {
      const int jn = j + 1;
      if (jn < jend) {
        fetch();                                             // rows of tile jn: in flight during the MFMAs below
        if (jn + 1 < jend) entries(jn + 1);
      }
      const double *tile = ds + buf * FIT_COV_TILE * ZS, *rt = rs + buf * FIT_COV_TILE;
      switch (wv) {
        case 0: fit_cov_tile<NBLK, 0, CMAX>(tile, rt, ZS, ar, ak, r_min, acc); break;
        case 1: fit_cov_tile<NBLK, 1, CMAX>(tile, rt, ZS, ar, ak, r_min, acc); break;
        case 2: fit_cov_tile<NBLK, 2, CMAX>(tile, rt, ZS, ar, ak, r_min, acc); break;
        default: fit_cov_tile<NBLK, 3, CMAX>(tile, rt, ZS, ar, ak, r_min, acc); break;
      }
      if (jn < jend) stage(buf ^ 1);                         // last read two tiles ago, before the previous barrier
      __syncthreads();
      buf ^= 1;
      j = jn;
    }
  }
  // lower-triangle blocks only (diagonal blocks in full); k_fit_reduce_sym mirrors them
  double *out = cpart + ((size_t)split * M + m) * D * D;
  switch (wv) {
    case 0: fit_cov_store<NBLK, 0, CMAX>(out, D, ar, ak, acc); break;
    case 1: fit_cov_store<NBLK, 1, CMAX>(out, D, ar, ak, acc); break;
    case 2: fit_cov_store<NBLK, 2, CMAX>(out, D, ar, ak, acc); break;
    default: fit_cov_store<NBLK, 3, CMAX>(out, D, ar, ak, acc); break;
  }
}

// sxx[m][i][j] = sum over splits of cpart[split][m][max(i,j)][min(i,j)]
__global__ void k_fit_reduce_sym(const double *__restrict__ cpart, int nsplit, int D, int M, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t len = (int64_t)M * D * D;
  if (e >= len) return;
  const int64_t m = e / ((int64_t)D * D);
  const int rem = (int)(e % ((int64_t)D * D));
  const int i = rem / D, j = rem % D;
  const int64_t src = m * D * D + (int64_t)max(i, j) * D + min(i, j);
  double s = 0.0;
  for (int c = 0; c < nsplit; ++c) s += cpart[(size_t)c * len + src];
  out[e] = s;
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

// frame splits per mixture in k_fit_logprob: one workgroup fits per CU (256 on the chip); two rounds,
// a multiple of 8 (the XCD-aware numbering), at least one 256-frame tile per workgroup
static int fit_lp_splits(int64_t n, int M) {
  const int64_t ntiles = (n + 255) / 256;
  int64_t s = (512 + M - 1) / M;
  s = (s + 7) & ~(int64_t)7;
  if (s > ntiles) s = ntiles;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

template <int NBLK>
static int fit_lp_launch(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int nsplit, const double *means,
                         const double *Zp, const double *cst, double *wlp) {
  const size_t lds = sizeof(double) * ((size_t)2 * NBLK * (NBLK + 1) * 64 + 16 * NBLK);
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_logprob<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_fit_logprob", hipLaunchKernelGGL(k_fit_logprob<NBLK>, dim3((unsigned)(M * nsplit)), dim3(FIT_LP_NT), lds, ctx->stream, X, n, D, M, nsplit, means, Zp, cst, wlp));
  return KWY_OK;
}

static int fit_cov_splits(int64_t n, int M);

// frames per workgroup of k_fit_sums: about 512 chunks, at least 512 frames each, a multiple of 4
static int fit_sum_rows(int64_t n) {
  int64_t r = (n + 511) / 512;
  if (r < 512) r = 512;
  return (int)((r + 3) & ~(int64_t)3);
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
  const int64_t nchunks = (n + fit_sum_rows(n) - 1) / fit_sum_rows(n);
  *bytes = (int64_t)(kwy_pad(sizeof(double) * fit_tri(D) * M) + kwy_pad(sizeof(double) * (size_t)M * D * D) +
                     kwy_pad(sizeof(double) * M) + kwy_pad(sizeof(double) * (size_t)nchunks * M * (D + 1)) +
                     kwy_pad(sizeof(double) * (size_t)fit_cov_splits(n, M) * M * D * D) + kwy_pad(64) + 4096);
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
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * nt * M) + kwy_pad(sizeof(double) * M)));
  double *Zp = kwy_arena<double>(ctx, nt * M);
  double *cst = kwy_arena<double>(ctx, M);
  if (!Zp || !cst) { ctx->err = "gmm_em_estep: scratch"; return KWY_ENOMEM; }
  KWY_HIP(hipMemsetAsync(status_out, 0, sizeof(int), ctx->stream));
  const size_t lds_prec = sizeof(double) * (nt + D);
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_prec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prec));
  hipLaunchKernelGGL(k_fit_prec, dim3(M), dim3(KWY_THREADS), lds_prec, ctx->stream, weights, covs, D, Zp, cst,
                     status_out);
  const int nsplit = fit_lp_splits(n, M);
  switch ((D + 15) / 16) {
    case 1: KWY_TRY(fit_lp_launch<1>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 2: KWY_TRY(fit_lp_launch<2>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 3: KWY_TRY(fit_lp_launch<3>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 4: KWY_TRY(fit_lp_launch<4>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 5: KWY_TRY(fit_lp_launch<5>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 6: KWY_TRY(fit_lp_launch<6>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 7: KWY_TRY(fit_lp_launch<7>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 8: KWY_TRY(fit_lp_launch<8>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    case 9: KWY_TRY(fit_lp_launch<9>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
    default: KWY_TRY(fit_lp_launch<10>(ctx, X, n, D, M, nsplit, means, Zp, cst, resp)); break;
  }
  hipLaunchKernelGGL(k_fit_resp, dim3((unsigned)((n + KWY_THREADS - 1) / KWY_THREADS)), dim3(KWY_THREADS), 0,
                     ctx->stream, resp, n, M, loglik_parts);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

template <int NBLK>
static void fit_sums_launch(kwy_ctx *ctx, const double *X, const double *resp, int64_t n, int D, int M, int rows,
                            int nchunks, double *part, const long long *gate, const int *labels) {
  KWY_PROF(ctx, "k_fit_sums", hipLaunchKernelGGL(k_fit_sums<NBLK>, dim3(nchunks, (M + 63) / 64), dim3(KWY_THREADS), 0, ctx->stream, X, resp, n, D, M, rows, part, gate, labels));
}

// stats[m][0] = sum_t r[t][m], stats[m][1+i] = sum_t r[t][m] x[t][i]   (local shard)
int kwy_fit_sums_gated(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp, double *stats,
                        const long long *gate, const int *labels) {
  KWY_TRY(fit_check(ctx, n, D, M));
  if (!X || (!resp && !labels) || !stats) { ctx->err = "gmm_em_sums: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int rows_per_chunk = fit_sum_rows(n);
  const int nchunks = (int)((n + rows_per_chunk - 1) / rows_per_chunk);
  const int64_t len = (int64_t)M * (D + 1);
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)nchunks * len)));
  double *part = kwy_arena<double>(ctx, (size_t)nchunks * len);
  if (!part) { ctx->err = "gmm_em_sums: scratch"; return KWY_ENOMEM; }
  switch ((D + 15) / 16) {
    case 1: fit_sums_launch<1>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 2: fit_sums_launch<2>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 3: fit_sums_launch<3>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 4: fit_sums_launch<4>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 5: fit_sums_launch<5>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 6: fit_sums_launch<6>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 7: fit_sums_launch<7>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 8: fit_sums_launch<8>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    case 9: fit_sums_launch<9>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
    default: fit_sums_launch<10>(ctx, X, resp, n, D, M, rows_per_chunk, nchunks, part, gate, labels); break;
  }
  hipLaunchKernelGGL(k_fit_reduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, part, nchunks, len,
                     stats, gate);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_gmm_em_sums_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                                   double *stats) {
  return kwy_fit_sums_gated(ctx, X, n, D, M, resp, stats, nullptr, nullptr);
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

// row splits per mixture in k_fit_cov: about two rounds of the 512 workgroups the chip holds, a multiple
// of 8 (the XCD-aware numbering), at least 512 frames per workgroup
static int fit_cov_splits(int64_t n, int M) {
  int64_t s = (1024 + M - 1) / M;
  s = (s + 7) & ~(int64_t)7;
  const int64_t most = (n + 511) / 512;
  if (s > most) s = most;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

template <int NBLK>
static int fit_cov_launch(kwy_ctx *ctx, const double *X, const double *resp, int64_t n, int D, int M,
                          const double *means, const double *stats, const int32_t *list, const int32_t *count,
                          int64_t ntiles, double *cpart, int nsplit) {
  const size_t lds = sizeof(double) * (2 * (size_t)FIT_COV_TILE * (16 * NBLK + 1) + 2 * FIT_COV_TILE);
  KWY_HIP(hipFuncSetAttribute((const void *)k_fit_cov<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_fit_cov", hipLaunchKernelGGL(k_fit_cov<NBLK>, dim3((unsigned)(M * nsplit)), dim3(KWY_THREADS), lds, ctx->stream, X, resp, n, D, M, nsplit, means, stats, list, count, ntiles, cpart));
  return KWY_OK;
}

// sxx[m] = sum_t r[t][m] (x_t - mu_m)(x_t - mu_m)'    (local shard; full D x D per mixture)
extern "C" int kwy_gmm_em_cov_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                                  const double *means, double *sxx) {
  return kwy_gmm_em_cov_stats_dev(ctx, X, n, D, M, resp, means, nullptr, sxx);
}

// ... with the (globally reduced) sums of kwy_gmm_em_sums_dev: frames whose responsibility is below 2^-70 of the
// mixture's mass are left out (see fit_cov_tile)
extern "C" int kwy_gmm_em_cov_stats_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp,
                                        const double *means, const double *stats, double *sxx) {
  KWY_TRY(fit_check(ctx, n, D, M));
  if (!X || !resp || !means || !sxx) { ctx->err = "gmm_em_cov: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int64_t len = (int64_t)M * D * D;
  const int nsplit = fit_cov_splits(n, M);
  const int64_t ntiles = (n + 3) / 4;            // list entries per mixture: groups of four frames
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)nsplit * len) + kwy_pad((size_t)M * ntiles) +
                               kwy_pad(sizeof(int32_t) * (size_t)M * ntiles) + kwy_pad(sizeof(int32_t) * (size_t)M)));
  double *cpart = kwy_arena<double>(ctx, (size_t)nsplit * len);
  unsigned char *act = (unsigned char *)kwy_arena_alloc(ctx, (size_t)M * ntiles);
  int32_t *list = kwy_arena<int32_t>(ctx, (size_t)M * ntiles);
  int32_t *count = kwy_arena<int32_t>(ctx, (size_t)M);
  if (!cpart || !act || !list || !count) { ctx->err = "gmm_em_cov: scratch"; return KWY_ENOMEM; }
  hipLaunchKernelGGL(k_fit_active, dim3((unsigned)((ntiles + FIT_ACT_GROUPS - 1) / FIT_ACT_GROUPS)), dim3(KWY_THREADS),
                     (size_t)M * FIT_ACT_GROUPS, ctx->stream, resp, n, D, M, stats, ntiles, act);
  hipLaunchKernelGGL(k_fit_active_list, dim3((unsigned)M), dim3(KWY_THREADS), 0, ctx->stream, act, ntiles, list, count);
  KWY_HIP(hipGetLastError());
  switch ((D + 15) / 16) {
    case 1: KWY_TRY(fit_cov_launch<1>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 2: KWY_TRY(fit_cov_launch<2>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 3: KWY_TRY(fit_cov_launch<3>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 4: KWY_TRY(fit_cov_launch<4>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 5: KWY_TRY(fit_cov_launch<5>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 6: KWY_TRY(fit_cov_launch<6>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 7: KWY_TRY(fit_cov_launch<7>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 8: KWY_TRY(fit_cov_launch<8>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    case 9: KWY_TRY(fit_cov_launch<9>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
    default: KWY_TRY(fit_cov_launch<10>(ctx, X, resp, n, D, M, means, stats, list, count, ntiles, cpart, nsplit)); break;
  }
  hipLaunchKernelGGL(k_fit_reduce_sym, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, cpart, nsplit, D,
                     M, sxx);
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
