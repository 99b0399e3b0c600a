// kwy_kmeans.hip -- k-means initialisation of the converter fit on gfx950.
//
// The reference trains with sklearn.mixture.GaussianMixture(init_params='kmeans')
// (kwiiyatta/converter/gmm.py:14-26): one KMeans run (k-means++ seeding, Lloyd iterations) gives the
// hard assignments of the first M-step.  These are the building blocks of that run, data-parallel
// over frames like the EM kernels (kwy_gmmfit.hip): every rank holds a shard of the rows, the
// driver (kwiiyatta_amd/converter/gmm_fit.py) exchanges the small per-step quantities -- shard
// totals, candidate rows, potentials, centroid sums and counts -- with all_reduce / all_gather.
//
//   k_km_colstats   column sums and sums of squares (mean of X; KMeans' tolerance = 1e-4 mean(var))
//   k_km_center     Xc = X - mean, row norms                      (KMeans centres its input)
//   k_km_pp_dist    squared distances of every row to <= 8 candidate rows, min with the running
//                   closest distance, potentials (sum over rows) per candidate      [k-means++]
//   k_km_chunk_sums / k_km_pick   searchsorted(cumsum(closest), rand * pot) without materialising
//                   the cumulative sum: chunk totals, then one workgroup walks to the hit
//   k_km_assign     nearest centre of every row: X C' on v_mfma_f64_16x16x4_f64, arg-min in registers
//   k_km_labels     labels, number of changed labels, one-hot responsibilities (EM's input)
//   k_km_update     new centres = sums / counts, squared centre shifts
// The centroid sums themselves are kwy_gmm_em_sums_dev with the one-hot responsibilities.
//
// All of it is HBM-bound except k_km_assign (2 n M D flop per Lloyd iteration).
#include <math.h>

#include "kwy_internal.hpp"

typedef double km_v4f64 __attribute__((ext_vector_type(4)));

#define KM_COLS 3          // columns per lane, one wavefront per row: D <= 192
#define KM_K16 12          // columns per lane, 16 lanes per row:      D <= 192
#define KM_MAXL 8          // candidates per k-means++ step (2 + ln(n_clusters): n_clusters <= 403)
#define KM_CHUNK 2048      // elements per workgroup in the chunk sums

template <class OP>
__device__ __forceinline__ double km_row_allreduce(double v, OP op) {   // over the 16 lanes of a DPP row
  v = op(v, kwy_dpp_f64<0x121>(v));   // row_ror:1
  v = op(v, kwy_dpp_f64<0x122>(v));
  v = op(v, kwy_dpp_f64<0x124>(v));
  v = op(v, kwy_dpp_f64<0x128>(v));
  return v;
}

// part[chunk][0][i] = sum_t (x[t][i] - shift[i]), part[chunk][1][i] = sum_t (x[t][i] - shift[i])^2
__global__ __launch_bounds__(KWY_THREADS) void k_km_colstats(const double *__restrict__ X, int64_t n, int D,
                                                            const double *__restrict__ shift, int rows_per_chunk,
                                                            double *__restrict__ part) {
  __shared__ double red[2 * 4 * 64 * KM_COLS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  double sh[KM_COLS], s[KM_COLS], q[KM_COLS];
#pragma unroll
  for (int c = 0; c < KM_COLS; ++c) {
    const int i = lane + 64 * c;
    sh[c] = (shift && i < D) ? shift[i] : 0.0;
    s[c] = q[c] = 0.0;
  }
  for (int64_t t = r0 + wv; t < r1; t += 4) {
#pragma unroll
    for (int c = 0; c < KM_COLS; ++c) {
      const int i = lane + 64 * c;
      if (i < D) {
        const double v = X[t * D + i] - sh[c];
        s[c] += v;
        q[c] += v * v;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < KM_COLS; ++c) {
    red[(wv * KM_COLS + c) * 64 + lane] = s[c];
    red[4 * 64 * KM_COLS + (wv * KM_COLS + c) * 64 + lane] = q[c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += KWY_THREADS) {
    const int which = e / D, i = e - which * D;
    const double *r = red + which * 4 * 64 * KM_COLS + (i >> 6) * 64 + (i & 63);
    part[(size_t)blockIdx.x * 2 * D + e] = ((r[0] + r[KM_COLS * 64]) + r[2 * KM_COLS * 64]) + r[3 * KM_COLS * 64];
  }
}

// out[e] = sum over chunks, in chunk order
__global__ void k_km_reduce(const double *__restrict__ part, int nchunks, int64_t len, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(size_t)c * len + e];
  out[e] = s;
}

// Xc = X - mean, xsq[t] = ||Xc[t]||^2 ; one wavefront per row
__global__ __launch_bounds__(KWY_THREADS) void k_km_center(const double *__restrict__ X, int64_t n, int D,
                                                          const double *__restrict__ mean, double *__restrict__ Xc,
                                                          double *__restrict__ xsq) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double mu[KM_COLS];
#pragma unroll
  for (int c = 0; c < KM_COLS; ++c) mu[c] = (lane + 64 * c < D) ? mean[lane + 64 * c] : 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wv; t < n; t += (int64_t)gridDim.x * 4) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < KM_COLS; ++c) {
      const int i = lane + 64 * c;
      if (i < D) {
        const double v = X[t * D + i] - mu[c];
        Xc[t * D + i] = v;
        acc += v * v;
      }
    }
    acc = kwy_wave_sum(acc);
    if (lane == 0) xsq[t] = acc;
  }
}

// k-means++ step.  cand: L x D candidate rows.  For every row t and candidate c (sklearn's
// euclidean_distances(cand, X, squared=True), then np.minimum with the closest distance so far):
//   d = max(0, (-2 <x_t, y_c> + |y_c|^2) + |x_t|^2),  newd[c][t] = min(closest[t], d)  (closest == NULL: d)
// part[block][c] = sum of newd[c][t] over the block's rows.  Sixteen lanes share a row.
__global__ __launch_bounds__(KWY_THREADS) void k_km_pp_dist(const double *__restrict__ Xc,
                                                           const double *__restrict__ xsq, int64_t n, int D,
                                                           const double *__restrict__ cand, int L,
                                                           const double *__restrict__ closest,
                                                           double *__restrict__ newd, double *__restrict__ part) {
  extern __shared__ double sm[];
  double *y = sm;                 // L x D
  double *ysq = y + (size_t)L * D;   // L
  double *red = ysq + KM_MAXL;    // 4 waves x 64 lanes
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ar = lane & 15, rs = lane >> 4;
  for (int e = tid; e < L * D; e += KWY_THREADS) y[e] = cand[e];
  __syncthreads();
  if (tid < L) {
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += y[tid * D + i] * y[tid * D + i];
    ysq[tid] = s;
  }
  __syncthreads();
  double pot = 0.0;               // lane (ar = c) of row slot rs: candidate c's potential over its rows
  const int64_t step = (int64_t)gridDim.x * 16;
  for (int64_t t0 = (int64_t)blockIdx.x * 16 + 4 * wv; t0 < n; t0 += step) {
    const int64_t t = t0 + rs;
    const bool live = t < n;
    const double *px = Xc + (live ? t : 0) * D;
    double x[KM_K16];
#pragma unroll
    for (int k = 0; k < KM_K16; ++k) {
      const int i = ar + 16 * k;
      x[k] = (live && i < D) ? px[i] : 0.0;
    }
    const double xs = live ? xsq[t] : 0.0;
    const double cl = (live && closest) ? closest[t] : INFINITY;
#pragma unroll
    for (int c = 0; c < KM_MAXL; ++c) {
      if (c < L) {   // uniform
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < KM_K16; ++k) {
          const int i = ar + 16 * k;
          if (i < D) dot += x[k] * y[c * D + i];
        }
        dot = km_row_allreduce(dot, [](double a, double b) { return a + b; });
        double d = (-2.0 * dot + ysq[c]) + xs;
        d = fmax(d, 0.0);
        d = fmin(cl, d);
        if (ar == c && live) {
          newd[(size_t)c * n + t] = d;
          pot += d;
        }
      }
    }
  }
  red[wv * 64 + lane] = pot;
  __syncthreads();
  if (tid < L) {
    double s = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int r = 0; r < 4; ++r) s += red[w * 64 + 16 * r + tid];
    part[(size_t)blockIdx.x * KM_MAXL + tid] = s;
  }
}

// csums[b] = sum of v[b*chunk .. (b+1)*chunk)
__global__ __launch_bounds__(KWY_THREADS) void k_km_chunk_sums(const double *__restrict__ v, int64_t n, int64_t chunk,
                                                              double *__restrict__ csums) {
  __shared__ double red[8];
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = min(n, b0 + chunk);
  double s = 0.0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += KWY_THREADS) s += v[i];
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) csums[blockIdx.x] = s;
}

// total[0] = sum of csums (single workgroup, fixed order per thread then tree)
__global__ __launch_bounds__(KWY_THREADS) void k_km_total(const double *__restrict__ csums, int nchunks,
                                                         double *__restrict__ total) {
  __shared__ double red[8];
  double s = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += KWY_THREADS) s += csums[i];
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) total[0] = s;
}

// np.searchsorted(lo + cumsum(v), vals[c]) restricted to this shard (single workgroup).
// idx[c] = local index, or -1 when the hit lies in another rank's shard:
//   mine  <=>  (first || vals[c] > lo) && (vals[c] <= hi || last);  beyond the end (last rank): n - 1
// lo / hi are the driver's cumulative sums of the all-gathered shard totals -- the SAME two numbers bound
// neighbouring shards on every rank, so a value on a boundary is owned by exactly one rank (with hi taken from this
// shard's own block sums, whose rounding differs from the gathered totals', two ranks or none could claim it).
__global__ __launch_bounds__(KWY_THREADS) void k_km_pick(const double *__restrict__ v, int64_t n, int64_t chunk,
                                                        const double *__restrict__ csums, int nchunks,
                                                        const double *__restrict__ lo_p,
                                                        const double *__restrict__ hi_p,
                                                        const double *__restrict__ vals, int L, int first, int last,
                                                        int64_t *__restrict__ idx) {
  extern __shared__ double sm[];
  double *pre = sm;                       // nchunks inclusive prefix
  double *buf = pre + nchunks;            // KM_CHUNK
  double *tot = buf + KM_CHUNK;           // KWY_THREADS
  __shared__ long long best;
  const int tid = threadIdx.x;
  for (int i = tid; i < nchunks; i += KWY_THREADS) pre[i] = csums[i];
  __syncthreads();
  kwy_block_cumsum(pre, nchunks, tot);
  const double lo = lo_p ? lo_p[0] : 0.0;
  const double hi = hi_p ? hi_p[0] : lo + pre[nchunks - 1];
  for (int c = 0; c < L; ++c) {
    const double val = vals[c] - lo;
    const bool mine = (first || vals[c] > lo) && (vals[c] <= hi || last);
    if (!mine) { if (tid == 0) idx[c] = -1; continue; }   // uniform
    if (tid == 0) best = (long long)nchunks;
    __syncthreads();
    long long mybest = nchunks;
    for (int i = tid; i < nchunks; i += KWY_THREADS)
      if (pre[i] >= val) { mybest = i; break; }
    if (mybest < nchunks) atomicMin(&best, mybest);
    __syncthreads();
    const long long kc = best;
    if (kc >= nchunks) { if (tid == 0) idx[c] = n - 1; __syncthreads(); continue; }   // beyond the end: clipped
    double base = kc > 0 ? pre[kc - 1] : 0.0;
    const int64_t b0 = kc * chunk, b1 = min(n, b0 + chunk);
    int64_t hit = b1 - 1;                  // rounding: the chunk total said "here", fall back to its last element
    for (int64_t s0 = b0; s0 < b1; s0 += KM_CHUNK) {
      const int len = (int)min((int64_t)KM_CHUNK, b1 - s0);
      __syncthreads();
      for (int i = tid; i < len; i += KWY_THREADS) buf[i] = v[s0 + i];
      if (tid == 0) best = (long long)len;
      __syncthreads();
      kwy_block_cumsum(buf, len, tot);
      const int per = (len + KWY_THREADS - 1) / KWY_THREADS;
      long long mine_i = len;
      for (int i = tid * per; i < min(len, (tid + 1) * per); ++i)
        if (base + buf[i] >= val) { mine_i = i; break; }
      if (mine_i < len) atomicMin(&best, mine_i);
      __syncthreads();
      if (best < len) { hit = s0 + best; break; }   // uniform
      base += buf[len - 1];
    }
    if (tid == 0) idx[c] = hit;
    __syncthreads();
  }
}

// Nearest centre: d(t, j) = |c_j|^2 - 2 <x_t, c_j>  (sklearn's lloyd_iter_chunked_dense), first minimum wins.
// The n x D by D x 64 product runs on v_mfma_f64_16x16x4_f64 like k_fit_logprob: a workgroup of eight
// wavefronts owns a group of <= 64 centres, expanded once into MFMA-fragment order in LDS (B[k l>>4][col l&15]);
// a wavefront takes 32 rows at a time straight from global memory into its A registers, so every B
// fragment it reads feeds two MFMAs; the arg-min over the centres is kept in registers (per lane over the
// centre blocks, then over the 16 lanes of a row by rotations).  pv / pi: [groups][n] best value / centre.
#define KM_AS_NT 512
template <int NBLK>
__global__ __launch_bounds__(KM_AS_NT) void k_km_assign(const double *__restrict__ X, int64_t n, int D, int M,
                                                       int nsplit, const double *__restrict__ centers,
                                                       double *__restrict__ pv, int *__restrict__ pi,
                                                       const long long *__restrict__ state) {
  // device-driven Lloyd loop (kwy_km_lloyd_dev): nothing once it has stopped; the centres of iteration state[1] are
  // buffer state[1] & 1 of the pair at `centers`
  if (state) {
    if (state[0]) return;
    centers += (size_t)(state[1] & 1) * M * D;
  }
  constexpr int KS = 4 * NBLK;
  extern __shared__ double sm[];
  double *cf = sm;                  // 4 x KS x 64
  double *cn = cf + 4 * KS * 64;    // 64: |c|^2, +inf beyond M
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ar = lane & 15, ak = lane >> 4;
  const int ngroups = (M + 63) / 64;
  const int g = blockIdx.x % ngroups, split = blockIdx.x / ngroups;
  const int m0 = 64 * g;
  for (int idx = tid; idx < 4 * KS * 64; idx += KM_AS_NT) {
    const int f = idx >> 6, l = idx & 63;
    const int mb = f / KS, ks = f - mb * KS;
    const int j = m0 + 16 * mb + (l & 15), k = 4 * ks + (l >> 4);
    cf[idx] = (j < M && k < D) ? centers[(size_t)j * D + k] : 0.0;
  }
  if (tid < 64) {
    const int j = m0 + tid;
    double s = INFINITY;
    if (j < M) {
      s = 0.0;
      for (int k = 0; k < D; ++k) s += centers[(size_t)j * D + k] * centers[(size_t)j * D + k];
    }
    cn[tid] = s;
  }
  __syncthreads();
  const int nmb = min(4, (M - m0 + 15) / 16);
  const int64_t ntiles = (n + 255) / 256;
  for (int64_t tile = split; tile < ntiles; tile += nsplit) {
    const int64_t t0 = tile * 256 + 32 * wv;
    if (t0 >= n) continue;
    const double *xa = X + min(t0 + ar, n - 1) * D, *xb = X + min(t0 + 16 + ar, n - 1) * D;
    double a0[KS], a1[KS];
#pragma unroll
    for (int ks = 0; ks < KS - 4; ++ks) {
      a0[ks] = xa[4 * ks + ak];
      a1[ks] = xb[4 * ks + ak];
    }
#pragma unroll
    for (int ks = KS - 4; ks < KS; ++ks) {   // columns beyond D meet zero fragments: clamp keeps the loads in bounds
      const int c = min(4 * ks + ak, D - 1);
      a0[ks] = xa[c];
      a1[ks] = xb[c];
    }
    double bv0[4], bv1[4];
    int bi0[4], bi1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv0[r] = bv1[r] = INFINITY; bi0[r] = bi1[r] = 0x7fffffff; }
    for (int mb = 0; mb < nmb; ++mb) {
      km_v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      const double *frag = cf + (size_t)mb * KS * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double b = frag[ks * 64];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b, acc1, 0, 0, 0);
      }
      const double c2 = cn[16 * mb + ar];
      const int j = m0 + 16 * mb + ar;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double d0 = c2 - 2.0 * acc0[r], d1 = c2 - 2.0 * acc1[r];
        if (d0 < bv0[r]) { bv0[r] = d0; bi0[r] = j; }
        if (d1 < bv1[r]) { bv1[r] = d1; bi1[r] = j; }
      }
    }
    // over the 16 lanes of the row: (value, centre) lexicographic minimum
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double ov0, ov1;
        int oi0, oi1;
        switch (s) {
          case 0: ov0 = kwy_dpp_f64<0x121>(bv0[r]); oi0 = (int)kwy_dpp_u32<0x121>((uint32_t)bi0[r]);
                  ov1 = kwy_dpp_f64<0x121>(bv1[r]); oi1 = (int)kwy_dpp_u32<0x121>((uint32_t)bi1[r]); break;
          case 1: ov0 = kwy_dpp_f64<0x122>(bv0[r]); oi0 = (int)kwy_dpp_u32<0x122>((uint32_t)bi0[r]);
                  ov1 = kwy_dpp_f64<0x122>(bv1[r]); oi1 = (int)kwy_dpp_u32<0x122>((uint32_t)bi1[r]); break;
          case 2: ov0 = kwy_dpp_f64<0x124>(bv0[r]); oi0 = (int)kwy_dpp_u32<0x124>((uint32_t)bi0[r]);
                  ov1 = kwy_dpp_f64<0x124>(bv1[r]); oi1 = (int)kwy_dpp_u32<0x124>((uint32_t)bi1[r]); break;
          default: ov0 = kwy_dpp_f64<0x128>(bv0[r]); oi0 = (int)kwy_dpp_u32<0x128>((uint32_t)bi0[r]);
                   ov1 = kwy_dpp_f64<0x128>(bv1[r]); oi1 = (int)kwy_dpp_u32<0x128>((uint32_t)bi1[r]); break;
        }
        if (ov0 < bv0[r] || (ov0 == bv0[r] && oi0 < bi0[r])) { bv0[r] = ov0; bi0[r] = oi0; }
        if (ov1 < bv1[r] || (ov1 == bv1[r] && oi1 < bi1[r])) { bv1[r] = ov1; bi1[r] = oi1; }
      }
    }
    if (ar == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t t = t0 + ak + 4 * r;
        if (t < n) { pv[(size_t)g * n + t] = bv0[r]; pi[(size_t)g * n + t] = bi0[r]; }
        if (t + 16 < n) { pv[(size_t)g * n + t + 16] = bv1[r]; pi[(size_t)g * n + t + 16] = bi1[r]; }
      }
    }
  }
}

// labels[t] = centre of the smallest pv over the groups (first group wins ties); changed += (label differs
// from the old one); resp[t][:] = one-hot (if resp).  Sixteen lanes share a row of resp.
__global__ __launch_bounds__(KWY_THREADS) void k_km_labels(const double *__restrict__ pv, const int *__restrict__ pi,
                                                          int ngroups, int64_t n, int M, int *__restrict__ labels,
                                                          double *__restrict__ resp,
                                                          unsigned long long *__restrict__ changed,
                                                          const long long *__restrict__ state = nullptr) {
  if (state && state[0]) return;
  const int c = threadIdx.x & 15, f = threadIdx.x >> 4;
  unsigned int ch = 0;
  for (int pass = 0; pass < KWY_THREADS / 16; ++pass) {
    const int64_t t = (int64_t)blockIdx.x * KWY_THREADS + 16 * pass + f;
    if (t >= n) continue;
    double bv = pv[t];
    int bi = pi[t];
    for (int g = 1; g < ngroups; ++g) {
      const double v = pv[(size_t)g * n + t];
      if (v < bv) { bv = v; bi = pi[(size_t)g * n + t]; }
    }
    if (c == 0) {
      if (labels[t] != bi) ++ch;
      labels[t] = bi;
    }
    if (resp)
      for (int m = c; m < M; m += 16) resp[t * M + m] = (m == bi) ? 1.0 : 0.0;
  }
  if (ch) atomicAdd(changed, (unsigned long long)ch);
}

// centres_new[j] = sums[j] * (1 / count[j]) (sklearn's _average_centers; a centre without rows keeps its
// place), shift2[j] = |new - old|^2.  stats: M x (1 + D) = [count, sums] (kwy_gmm_em_sums_dev layout).
__global__ __launch_bounds__(KWY_THREADS) void k_km_update(const double *__restrict__ stats,
                                                          const double *__restrict__ cold, int M, int D,
                                                          double *__restrict__ cnew, double *__restrict__ shift2) {
  __shared__ double red[8];
  const int j = blockIdx.x;
  const double cnt = stats[(size_t)j * (D + 1)];
  const double inv = cnt > 0.0 ? 1.0 / cnt : 0.0;
  double s = 0.0;
  for (int i = threadIdx.x; i < D; i += KWY_THREADS) {
    const double o = cold[(size_t)j * D + i];
    const double v = cnt > 0.0 ? stats[(size_t)j * (D + 1) + 1 + i] * inv : o;
    cnew[(size_t)j * D + i] = v;
    s += (v - o) * (v - o);
  }
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) shift2[j] = s;
}

// ---- C ABI ------------------------------------------------------------------------------------------
static int km_check(kwy_ctx *ctx, int64_t n, int D) {
  if (!ctx) return KWY_EINVAL;
  if (n <= 0 || D <= 0 || D > 160) { ctx->err = "kmeans: need n > 0 and 0 < D <= 160"; return KWY_EINVAL; }
  return KWY_OK;
}

static int km_rows_per_chunk(int64_t n) {
  int64_t r = (n + 511) / 512;
  if (r < 256) r = 256;
  return (int)r;
}

// elements per chunk of the cumulative-sum search: at most 4096 chunks, a multiple of KM_CHUNK
static int64_t km_chunk(int64_t n) {
  const int64_t mult = (n + (int64_t)KM_CHUNK * 4096 - 1) / ((int64_t)KM_CHUNK * 4096);
  return KM_CHUNK * (mult < 1 ? 1 : mult);
}

extern "C" int64_t kwy_km_chunks(int64_t n) {
  if (n <= 0) return 0;
  return (n + km_chunk(n) - 1) / km_chunk(n);
}

// out[0..D) = sum_t (X[t] - shift), out[D..2D) = sum_t (X[t] - shift)^2     (shift may be NULL)
extern "C" int kwy_km_colstats_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *shift,
                                   double *out) {
  KWY_TRY(km_check(ctx, n, D));
  if (!X || !out) { ctx->err = "km_colstats: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int rows = km_rows_per_chunk(n);
  const int nchunks = (int)((n + rows - 1) / rows);
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)nchunks * 2 * D)));
  double *part = kwy_arena<double>(ctx, (size_t)nchunks * 2 * D);
  if (!part) { ctx->err = "km_colstats: scratch"; return KWY_ENOMEM; }
  hipLaunchKernelGGL(k_km_colstats, dim3(nchunks), dim3(KWY_THREADS), 0, ctx->stream, X, n, D, shift, rows, part);
  hipLaunchKernelGGL(k_km_reduce, dim3((2 * D + 255) / 256), dim3(256), 0, ctx->stream, part, nchunks,
                     (int64_t)2 * D, out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// Xc = X - mean (n x D), xsq[t] = |Xc[t]|^2
extern "C" int kwy_km_center_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *mean, double *Xc,
                                 double *xsq) {
  KWY_TRY(km_check(ctx, n, D));
  if (!X || !mean || !Xc || !xsq) { ctx->err = "km_center: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const unsigned grid = (unsigned)min((int64_t)4096, (n + 3) / 4);
  hipLaunchKernelGGL(k_km_center, dim3(grid), dim3(KWY_THREADS), 0, ctx->stream, X, n, D, mean, Xc, xsq);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// newd: L x n, pots: L (sums over this shard).  closest may be NULL (first centre).
extern "C" int kwy_km_pp_dist_dev(kwy_ctx *ctx, const double *Xc, const double *xsq, int64_t n, int D,
                                  const double *cand, int L, const double *closest, double *newd, double *pots) {
  KWY_TRY(km_check(ctx, n, D));
  if (!Xc || !xsq || !cand || !newd || !pots || L < 1 || L > KM_MAXL) {
    ctx->err = "km_pp_dist: null pointer or more than 8 candidates";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int grid = (int)min((int64_t)1024, (n + 15) / 16);
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)grid * KM_MAXL)));
  double *part = kwy_arena<double>(ctx, (size_t)grid * KM_MAXL);
  if (!part) { ctx->err = "km_pp_dist: scratch"; return KWY_ENOMEM; }
  KWY_HIP(hipMemsetAsync(part, 0, sizeof(double) * (size_t)grid * KM_MAXL, ctx->stream));
  const size_t lds = sizeof(double) * ((size_t)L * D + KM_MAXL + 256);
  KWY_PROF(ctx, "k_km_pp_dist", hipLaunchKernelGGL(k_km_pp_dist, dim3(grid), dim3(KWY_THREADS), lds, ctx->stream, Xc, xsq, n, D, cand, L, closest, newd, part));
  hipLaunchKernelGGL(k_km_reduce, dim3(1), dim3(64), 0, ctx->stream, part, grid, (int64_t)KM_MAXL, pots);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// csums: kwy_km_chunks(n) doubles (chunk totals of v), total[0] = their sum
extern "C" int kwy_km_pp_total_dev(kwy_ctx *ctx, const double *v, int64_t n, double *csums, double *total) {
  if (!ctx) return KWY_EINVAL;
  if (!v || !csums || !total || n <= 0) { ctx->err = "km_pp_total: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int nchunks = (int)kwy_km_chunks(n);
  hipLaunchKernelGGL(k_km_chunk_sums, dim3(nchunks), dim3(KWY_THREADS), 0, ctx->stream, v, n, km_chunk(n), csums);
  hipLaunchKernelGGL(k_km_total, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, csums, nchunks, total);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// idx[c] (int64, c < L): np.searchsorted(lo + cumsum(v), vals[c]) as a local index, -1 if it lies in another
// shard (first / last: this is the first / last shard in the global row order).  lo / hi: device scalars, the
// cumulated totals of the shards before this one / up to and including this one (NULL: 0 / lo + this shard's total).
extern "C" int kwy_km_pp_pick_dev(kwy_ctx *ctx, const double *v, int64_t n, const double *csums, const double *lo,
                                  const double *hi, const double *vals, int L, int first, int last, int64_t *idx) {
  if (!ctx) return KWY_EINVAL;
  if (!v || !csums || !vals || !idx || n <= 0 || L < 1 || L > KM_MAXL) {
    ctx->err = "km_pp_pick: null pointer or more than 8 values";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int nchunks = (int)kwy_km_chunks(n);
  const size_t lds = sizeof(double) * ((size_t)nchunks + KM_CHUNK + KWY_THREADS);
  KWY_HIP(hipFuncSetAttribute((const void *)k_km_pick, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_km_pick, dim3(1), dim3(KWY_THREADS), lds, ctx->stream, v, n, km_chunk(n), csums, nchunks, lo,
                     hi, vals, L, first, last, idx);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

template <int NBLK>
static int km_assign_launch(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int nsplit, const double *centers,
                            double *pv, int *pi, const long long *state) {
  const size_t lds = sizeof(double) * ((size_t)4 * 4 * NBLK * 64 + 64);
  KWY_HIP(hipFuncSetAttribute((const void *)k_km_assign<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ngroups = (M + 63) / 64;
  KWY_PROF(ctx, "k_km_assign", hipLaunchKernelGGL(k_km_assign<NBLK>, dim3((unsigned)(ngroups * nsplit)), dim3(KM_AS_NT), lds, ctx->stream, X, n, D, M, nsplit, centers, pv, pi, state));
  return KWY_OK;
}

// labels (int32, n; in: previous labels, out: nearest centre), resp (n x M one-hot, may be NULL),
// changed (device uint64: number of rows whose label changed)
// state: NULL, or the control words of kwy_km_lloyd_dev (centers is then the PAIR of centre buffers)
static int km_assign_core(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *centers, int M,
                          int *labels, double *resp, unsigned long long *changed, const long long *state) {
  KWY_TRY(km_check(ctx, n, D));
  if (!X || !centers || !labels || !changed || M < 1 || M > 256) {
    ctx->err = "km_assign: null pointer or M outside 1..256";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int ngroups = (M + 63) / 64;
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * (size_t)ngroups * n) + kwy_pad(sizeof(int) * (size_t)ngroups * n)));
  double *pv = kwy_arena<double>(ctx, (size_t)ngroups * n);
  int *pi = kwy_arena<int>(ctx, (size_t)ngroups * n);
  if (!pv || !pi) { ctx->err = "km_assign: scratch"; return KWY_ENOMEM; }
  const int64_t ntiles = (n + 255) / 256;
  int nsplit = (int)min(ntiles, (int64_t)((512 + ngroups - 1) / ngroups));
  if (nsplit < 1) nsplit = 1;
  switch ((D + 15) / 16) {
    case 1: KWY_TRY(km_assign_launch<1>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 2: KWY_TRY(km_assign_launch<2>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 3: KWY_TRY(km_assign_launch<3>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 4: KWY_TRY(km_assign_launch<4>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 5: KWY_TRY(km_assign_launch<5>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 6: KWY_TRY(km_assign_launch<6>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 7: KWY_TRY(km_assign_launch<7>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 8: KWY_TRY(km_assign_launch<8>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    case 9: KWY_TRY(km_assign_launch<9>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
    default: KWY_TRY(km_assign_launch<10>(ctx, X, n, D, M, nsplit, centers, pv, pi, state)); break;
  }
  KWY_HIP(hipMemsetAsync(changed, 0, sizeof(unsigned long long), ctx->stream));   // (a stopped loop has its count in state[2])
  hipLaunchKernelGGL(k_km_labels, dim3((unsigned)((n + KWY_THREADS - 1) / KWY_THREADS)), dim3(KWY_THREADS), 0,
                     ctx->stream, pv, pi, ngroups, n, M, labels, resp, changed, state);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_km_assign_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, const double *centers, int M,
                                 int *labels, double *resp, unsigned long long *changed) {
  return km_assign_core(ctx, X, n, D, centers, M, labels, resp, changed, nullptr);
}

// stats: M x (1 + D) globally reduced [count, sums]; centers_new[j] = sums / count; shift2[j] = |new - old|^2
extern "C" int kwy_km_update_dev(kwy_ctx *ctx, const double *stats, const double *centers_old, int M, int D,
                                 double *centers_new, double *shift2) {
  KWY_TRY(km_check(ctx, 1, D));
  if (!stats || !centers_old || !centers_new || !shift2 || M < 1) { ctx->err = "km_update: null pointer"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_km_update, dim3(M), dim3(KWY_THREADS), 0, ctx->stream, stats, centers_old, M, D, centers_new,
                     shift2);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// ---- the Lloyd loop without the host in it ------------------------------------------------------------------------
// sklearn's _kmeans_single_lloyd asks after every iteration whether to stop (labels unchanged / centre shift below
// the tolerance), which costs a device -> host round trip per iteration: about 0.55 ms next to 0.75 ms of kernels at
// 4.4e5 x 144, M = 64.  Here the DEVICE takes that decision (k_km_decide) and the kernels of the iterations enqueued
// behind the one that stopped return at once, so the host enqueues a batch of iterations and reads the control words
// once per batch.  state (int64[4]): [0] 0 = running, 1 = labels unchanged (strict convergence), 2 = centre shift
// <= abs_tol, 3 = an EMPTY CLUSTER (sklearn relocates it before the update: the iteration is left after its sums --
// labels, resp and stats are this iteration's, the centres are not updated -- for the host to finish), 4 = max_iter
// reached; [1] finished iterations (the centres of iteration i are buffer i & 1 of `centers2`); [2] changed labels
// of the last assignment.  log: 2 doubles per finished iteration (changed labels, centre shift).

// centres_new[j] = sums / count, shift2[j] = |new - old|^2 as k_km_update -- unless the loop has stopped or a
// cluster is empty
__global__ __launch_bounds__(KWY_THREADS) void k_km_update_gated(const double *__restrict__ stats, double *__restrict__ c2,
                                                                int M, int D, double *__restrict__ shift2,
                                                                const long long *__restrict__ state) {
  __shared__ double red[8];
  __shared__ int empty;
  if (state[0]) return;
  if (threadIdx.x == 0) empty = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < M; j += KWY_THREADS)
    if (stats[(size_t)j * (D + 1)] == 0.0) empty = 1;
  __syncthreads();
  if (empty) return;
  const int cur = (int)(state[1] & 1);
  const double *cold = c2 + (size_t)cur * M * D;
  double *cnew = c2 + (size_t)(cur ^ 1) * M * D;
  const int j = blockIdx.x;
  const double cnt = stats[(size_t)j * (D + 1)];
  const double inv = 1.0 / cnt;
  double s = 0.0;
  for (int i = threadIdx.x; i < D; i += KWY_THREADS) {
    const double o = cold[(size_t)j * D + i];
    const double v = stats[(size_t)j * (D + 1) + 1 + i] * inv;
    cnew[(size_t)j * D + i] = v;
    s += (v - o) * (v - o);
  }
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) shift2[j] = s;
}

// numpy's sum of n <= 256 contiguous doubles (pairwise_sum: a plain loop below eight elements; up to 128 eight running
// sums over i mod 8, combined pairwise, then the tail; longer vectors split at n/2 rounded down to a multiple of 8)
__device__ inline double km_np_sum_block(const double *a, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i];
    return s;
  }
  double r[8];
  for (int q = 0; q < 8; ++q) r[q] = a[q];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int q = 0; q < 8; ++q) r[q] += a[i + q];
  double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s += a[i];
  return s;
}
__device__ inline double km_np_sum_half(const double *a, int n) {      // n <= 256 / 2 + 8
  if (n <= 128) return km_np_sum_block(a, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return km_np_sum_block(a, n2) + km_np_sum_block(a + n2, n - n2);
}
__device__ inline double km_np_sum(const double *a, int n) {           // n <= 256
  if (n <= 128) return km_np_sum_block(a, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return km_np_sum_half(a, n2) + km_np_sum_half(a + n2, n - n2);
}

// one thread: the decision after an iteration, as sklearn's loop takes it: labels unchanged first, then
// `(center_shift ** 2).sum() <= tol` with numpy's summation order
__global__ void k_km_decide(const double *__restrict__ stats, const unsigned long long *__restrict__ changed,
                            const double *__restrict__ shift2, int M, int D, double abs_tol, long long max_iter,
                            long long *__restrict__ state, double *__restrict__ log) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || state[0]) return;
  state[2] = (long long)changed[0];
  for (int j = 0; j < M; ++j)
    if (stats[(size_t)j * (D + 1)] == 0.0) { state[0] = 3; return; }
  const double tot = km_np_sum(shift2, M);
  const long long it = state[1] + 1;
  log[2 * (it - 1)] = (double)changed[0];
  log[2 * (it - 1) + 1] = tot;
  state[1] = it;
  if (changed[0] == 0ull) state[0] = 1;
  else if (tot <= abs_tol) state[0] = 2;
  else if (it >= max_iter) state[0] = 4;
}

extern "C" int kwy_km_lloyd_dev(kwy_ctx *ctx, const double *Xc, int64_t n, int D, double *centers2, int M,
                                int *labels, double *resp, unsigned long long *changed, double *stats,
                                double *shift2, double abs_tol, int iterations, int64_t max_iter, long long *state,
                                double *log) {
  KWY_TRY(km_check(ctx, n, D));
  (void)resp;    // (the one-hot rows are not kept up to date by the loop: kwy_km_onehot_dev after it)
  if (!Xc || !centers2 || !labels || !changed || !stats || !shift2 || !state || !log || M < 1 || M > 256 ||
      iterations < 1 || max_iter < 1) {
    ctx->err = "km_lloyd: null pointer, M outside 1..256 or no iterations";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  for (int it = 0; it < iterations; ++it) {
    KWY_TRY(km_assign_core(ctx, Xc, n, D, centers2, M, labels, nullptr, changed, state));
    KWY_TRY(kwy_fit_sums_gated(ctx, Xc, n, D, M, nullptr, stats, state, labels));
    hipLaunchKernelGGL(k_km_update_gated, dim3(M), dim3(KWY_THREADS), 0, ctx->stream, stats, centers2, M, D, shift2,
                       state);
    hipLaunchKernelGGL(k_km_decide, dim3(1), dim3(64), 0, ctx->stream, stats, changed, shift2, M, D, abs_tol,
                       (long long)max_iter, state, log);
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// resp[t][:] = the one-hot row of labels[t] (what kwy_km_assign_dev writes beside the labels)
__global__ __launch_bounds__(KWY_THREADS) void k_km_onehot(const int *__restrict__ labels, int64_t n, int M,
                                                          double *__restrict__ resp) {
  const int c = threadIdx.x & 15, f = threadIdx.x >> 4;
  for (int pass = 0; pass < KWY_THREADS / 16; ++pass) {
    const int64_t t = (int64_t)blockIdx.x * KWY_THREADS + 16 * pass + f;
    if (t >= n) continue;
    const int bi = labels[t];
    for (int m = c; m < M; m += 16) resp[t * M + m] = (m == bi) ? 1.0 : 0.0;
  }
}

extern "C" int kwy_km_onehot_dev(kwy_ctx *ctx, const int *labels, int64_t n, int M, double *resp) {
  if (!ctx) return KWY_EINVAL;
  if (!labels || !resp || n <= 0 || M < 1) { ctx->err = "km_onehot: null pointer or empty"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_km_onehot, dim3((unsigned)((n + KWY_THREADS - 1) / KWY_THREADS)), dim3(KWY_THREADS), 0,
                     ctx->stream, labels, n, M, resp);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}
