// kwy_dtw.hip -- FastDTW (multi-resolution, windowed dynamic time warping) on gfx950.
//
// Replaces fastdtw.fastdtw(x, y, radius, dist=2) (reference call site
// kwiiyatta/vocoder/align.py:71; fastdtw 0.3.2).  The level recursion is unrolled
// on the host (sizes depend only on Tx, Ty, radius); everything else stays on
// the device with no host synchronisation:
//
//   k_dtw_halve    coarsen a series: (x[2i] + x[2i+1]) / 2
//   k_dtw_window   per-row column window from the coarser level's path
//                  (monotone path => two binary searches per row, no atomics)
//   k_dtw_dist     Euclidean frame distances for every cell of the band (parallel)
//   k_dtw_dp       ONE wavefront: rows are processed in strips of 64, lane = row,
//                  skewed so that lane l works on column j - l; the three
//                  predecessors arrive by wave shuffles (DPP), the strip boundary
//                  row goes through LDS; then the back-trace, strip by strip,
//                  over predecessor codes staged in LDS.
//
// Tie-breaking follows fastdtw's pure-Python min(): (i-1,j), (i,j-1), (i-1,j-1).
#include <math.h>

#include <vector>

#include "kwy_internal.hpp"

#define DTW_PAD 128          // slack cells before/after the band storage
#define DTW_BT_BYTES 49152   // LDS budget for staged predecessor codes
#define DTW_BT_ROWS 512      // rows per back-trace group

__global__ void k_dtw_halve(const double *__restrict__ in, int n_out, int dim, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n_out * dim) return;
  const int i = (int)(e / dim), k = (int)(e % dim);
  out[e] = (in[(int64_t)(2 * i) * dim + k] + in[(int64_t)(2 * i + 1) * dim + k]) / 2;
}

// window rows.  cpath == nullptr: full window (base case).  Otherwise cpath is the
// coarser level's path (cn = *cpath_len cells, both coordinates non-decreasing).
__global__ void k_dtw_window(const int32_t *__restrict__ cpath, const int64_t *__restrict__ cpath_len,
                             int radius, int len_x, int len_y, int32_t *__restrict__ lo,
                             int32_t *__restrict__ hi, uint32_t *__restrict__ width) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len_x) return;
  int l = 0, h = len_y - 1;
  if (cpath) {
    const int cn = (int)*cpath_len;
    if (cn <= 0) { lo[i] = 0; hi[i] = len_y - 1; width[i] = (uint32_t)len_y; return; }
    const int a = i / 2;
    // first cell with pi >= a - radius  -> smallest pj among cells within +-radius rows
    int b0 = 0, b1 = cn;
    while (b0 < b1) { int mid = (b0 + b1) >> 1; if (cpath[2 * mid] >= a - radius) b1 = mid; else b0 = mid + 1; }
    const int first = min(b0, cn - 1);
    // last cell with pi <= a + radius
    b0 = 0; b1 = cn;
    while (b0 < b1) { int mid = (b0 + b1) >> 1; if (cpath[2 * mid] > a + radius) b1 = mid; else b0 = mid + 1; }
    const int last = max(b0 - 1, 0);
    l = 2 * (cpath[2 * first + 1] - radius);
    h = 2 * (cpath[2 * last + 1] + radius) + 1;
    if (l < 0) l = 0;
    if (h > len_y - 1) h = len_y - 1;
  }
  lo[i] = l;
  hi[i] = h;
  width[i] = (uint32_t)(h - l + 1);
}

// dist[off[i] + j - lo[i]] = || x_i - y_j ||_2   (sequential sum over the dimensions)
__global__ __launch_bounds__(KWY_THREADS) void k_dtw_dist(const double *__restrict__ x,
                                                         const double *__restrict__ y, int dim,
                                                         const int32_t *__restrict__ lo,
                                                         const int32_t *__restrict__ hi,
                                                         const uint64_t *__restrict__ off, uint64_t cap,
                                                         double *__restrict__ dist, int *__restrict__ status) {
  extern __shared__ double xs[];
  const int i = blockIdx.x;
  if (off[i] + (uint64_t)(hi[i] - lo[i] + 1) > cap) { if (threadIdx.x == 0) atomicExch(status, 1); return; }
  for (int k = threadIdx.x; k < dim; k += KWY_THREADS) xs[k] = x[(int64_t)i * dim + k];
  __syncthreads();
  double *row = dist + DTW_PAD + off[i];
  const int l = lo[i], h = hi[i];
  for (int j = l + threadIdx.x; j <= h; j += KWY_THREADS) {
    const double *yr = y + (int64_t)j * dim;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) { double d = xs[k] - yr[k]; s += d * d; }
    row[j - l] = sqrt(s);
  }
}

// lane l <- lane l-1 across the whole wavefront (DPP wave_shr:1, no LDS round trip);
// lane 0 keeps its own value.
__device__ __forceinline__ double dtw_wave_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// value of lane `idx` (wave-uniform index) broadcast to all lanes (v_readlane, scalar path)
__device__ __forceinline__ double dtw_readlane(double v, int idx) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), idx);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), idx);
  return __hiloint2double(hi, lo);
}

// Packed predecessor codes: 2 bits per cell, 32 cells per 64-bit word; row i owns the
// words starting at (off[i] >> 5) + i (rows never share a word).
__device__ __forceinline__ uint64_t dtw_word_base(const uint64_t *__restrict__ off, int i) {
  return (off[i] >> 5) + (uint64_t)i;
}

// The DP + back-trace of one level, one wavefront.
template <bool BND_LDS>
__global__ __launch_bounds__(64) void k_dtw_dp(int len_x, int len_y, const int32_t *__restrict__ lo,
                                              const int32_t *__restrict__ hi,
                                              const uint64_t *__restrict__ off, uint64_t cap,
                                              const double *__restrict__ dist,
                                              unsigned long long *__restrict__ predw,
                                              double *__restrict__ bnd_global /* 2 x (len_y+2) or null */,
                                              int32_t *__restrict__ path, int32_t *__restrict__ rev,
                                              int64_t *__restrict__ path_len, double *__restrict__ out_dist,
                                              const int *__restrict__ status) {
  extern __shared__ unsigned char bt[];  // DTW_BT_BYTES [+ boundary rows]
  __shared__ int s_i, s_j, s_n;
  const int lane = threadIdx.x;
  if (*status != 0) { if (lane == 0) { *path_len = 0; *out_dist = NAN; } return; }
  const double INF = INFINITY;
  constexpr bool bnd_lds = BND_LDS;
  // boundary rows are indexed by j + 1 (entry 0 is column -1)
  double *const lds_rows = (double *)(bt + DTW_BT_BYTES);
#define B_PREV(ix) (BND_LDS ? lds_rows[o_prev + (ix)] : bnd_global[o_prev + (ix)])
#define B_NEXT(ix) (BND_LDS ? lds_rows[o_next + (ix)] : bnd_global[o_next + (ix)])
  int o_prev = 0, o_next = len_y + 2;
  for (int j = lane; j < len_y + 2; j += 64) B_PREV(j) = INF;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) B_PREV(0) = 0.0;  // D[-1][-1] = 0: the origin of the recurrence
  double last_val = INF;
  for (int i0 = 0; i0 < len_x; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < len_x;
    const int rl = valid ? lo[i] : 0, rh = valid ? hi[i] : -1;
    const int rw = rh - rl;  // last valid index of the row
    const int rwc = rw > 0 ? rw : 0;
    const int ilast = min(i0 + 63, len_x - 1);
    const int jmin = lo[i0], jmax = hi[ilast];
    for (int j = lane; j < len_y + 2; j += 64) B_NEXT(j) = INF;
    __builtin_amdgcn_wave_barrier();
    if (bnd_lds) __threadfence_block(); else __threadfence();
    const double *drow = dist + DTW_PAD + (valid ? off[i] : 0);
    unsigned long long *pwrow = predw + (valid ? dtw_word_base(off, i) : 0);
    double v1 = INF, v2 = INF;  // this lane's values at the two previous steps
    const int nsteps = (jmax - jmin + 1) + 63;
    const int shift = lane + rl - jmin;  // this lane's row index at step s is s - shift
    unsigned long long pw = 0ull;        // predecessor codes of the current 32-cell word
    double up0_prev = B_PREV(jmin);      // lane 0: D[i0-1][jmin-1] (column j-1 of the first step)
    // software prefetch of the lane's distances, 8 steps per block
    double curd[8], nxtd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      curd[u] = drow[min(max(u - shift, 0), rwc)];
    }
    for (int c0 = 0; c0 < nsteps; c0 += 64) {
      // the boundary row values lane 0 will need in the next 64 steps: one LDS read per lane
      const int jb = jmin + c0 + lane;
      const double bchunk = (jb >= -1 && jb <= len_y) ? B_PREV(jb + 1) : INF;
      for (int s0 = c0; s0 < min(c0 + 64, nsteps); s0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          nxtd[u] = drow[min(max(s0 + 8 + u - shift, 0), rwc)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s = s0 + u;
          const int j = jmin + s - lane;
          const int pos = s - shift;
          const bool act = pos >= 0 && pos <= rw;
          double up = dtw_wave_shr1(v1), dg = dtw_wave_shr1(v2);
          const double bup = dtw_readlane(bchunk, (s - c0) & 63);
          if (lane == 0) { up = bup; dg = up0_prev; }
          up0_prev = bup;
          double cur = INF;
          if (act) {
            const double dt = curd[u];
            const double c0v = up + dt, c1v = v1 + dt, c2v = dg + dt;
            double best = c0v; unsigned long long pb = 0ull;
            if (c1v < best) { best = c1v; pb = 1ull; }
            if (c2v < best) { best = c2v; pb = 2ull; }
            cur = best;
            pw |= pb << (2 * (pos & 31));
            if ((pos & 31) == 31 || pos == rw) { pwrow[pos >> 5] = pw; pw = 0ull; }
            if (i == ilast) B_NEXT(j + 1) = cur;
            if (i == len_x - 1 && j == len_y - 1) last_val = cur;
          }
          v2 = v1;
          v1 = cur;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) curd[u] = nxtd[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (bnd_lds) __threadfence_block(); else __threadfence();
    { int t = o_prev; o_prev = o_next; o_next = t; }
  }
#undef B_PREV
#undef B_NEXT
  // broadcast D[len_x-1][len_y-1] (held by the lane of the last row)
  {
    const int owner = (len_x - 1) & 63;
    last_val = __shfl(last_val, owner);
    if (lane == 0) *out_dist = last_val;
  }
  __threadfence();  // predecessor codes written above are read back below through global memory

  // ---- back-trace, staged through LDS in groups of rows (codes AND row descriptors,
  //      so that the serial walk of lane 0 never waits on global memory)
  unsigned long long *btw = (unsigned long long *)bt;
  const uint64_t bt_words = DTW_BT_BYTES / 8;
  __shared__ int g_lo[DTW_BT_ROWS], g_hi[DTW_BT_ROWS];
  __shared__ unsigned int g_wb[DTW_BT_ROWS];
  int ci = len_x - 1, cj = len_y - 1, n = 0;
  while (ci >= 0) {
    // rows [r0, ci]: at most DTW_BT_ROWS rows whose code words fit in the LDS budget
    int r0 = ci;
    const uint64_t wend = dtw_word_base(off, ci) + (uint64_t)((hi[ci] - lo[ci]) >> 5) + 1;
    while (r0 > 0 && ci - (r0 - 1) < DTW_BT_ROWS && wend - dtw_word_base(off, r0 - 1) <= bt_words) --r0;
    const uint64_t wbase = dtw_word_base(off, r0);
    const uint64_t nw = wend - wbase;
    const bool staged = nw <= bt_words;
    if (staged)
      for (uint64_t b = lane; b < nw; b += 64) btw[b] = predw[wbase + b];
    for (int r = r0 + lane; r <= ci; r += 64) {
      g_lo[r - r0] = lo[r];
      g_hi[r - r0] = hi[r];
      g_wb[r - r0] = (unsigned int)(dtw_word_base(off, r) - wbase);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane == 0) {
      int i = ci, j = cj, m = n;
      while (i >= r0) {
        rev[2 * m] = i; rev[2 * m + 1] = j; ++m;
        if (i == 0 && j == 0) { i = -1; break; }
        unsigned int pb = 0;
        const int l = g_lo[i - r0];
        if (j >= l && j <= g_hi[i - r0]) {
          const unsigned int w = g_wb[i - r0] + (unsigned int)((j - l) >> 5);
          const unsigned long long word = staged ? btw[w] : predw[wbase + w];
          pb = (unsigned int)(word >> (2 * ((j - l) & 31))) & 3u;
        }
        if (pb == 0) --i; else if (pb == 1) --j; else { --i; --j; }
        if (j < 0) { i = -1; break; }
      }
      s_i = i; s_j = j; s_n = m;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    ci = s_i; cj = s_j; n = s_n;
    __builtin_amdgcn_wave_barrier();
  }
  for (int k = lane; k < n; k += 64) {
    path[2 * k] = rev[2 * (n - 1 - k)];
    path[2 * k + 1] = rev[2 * (n - 1 - k) + 1];
  }
  if (lane == 0) *path_len = n;
}

// ---- host side -----------------------------------------------------------------------------
struct dtw_level { int len_x, len_y; };

static uint64_t dtw_cap(int len_x, int len_y, int radius, bool full) {
  uint64_t worst = (uint64_t)len_x * (uint64_t)len_y;
  if (full) return worst;
  uint64_t est = (uint64_t)len_x * (uint64_t)(8 * radius + 64) * 2;
  return est < worst ? est : worst;
}

static size_t dtw_scratch_bytes(int64_t Tx, int64_t Ty, int dim, int radius, bool full) {
  size_t tot = 0;
  int lx = (int)Tx, ly = (int)Ty;
  // coarsened series
  while (!(lx < radius + 2 || ly < radius + 2)) {
    lx /= 2; ly /= 2;
    tot += kwy_pad(sizeof(double) * (size_t)lx * dim) + kwy_pad(sizeof(double) * (size_t)ly * dim);
  }
  uint64_t cap = dtw_cap((int)Tx, (int)Ty, radius, full);
  tot += kwy_pad(sizeof(double) * (cap + 2 * DTW_PAD)) + kwy_pad(8 * (cap / 32 + Tx + 64));
  tot += 2 * kwy_pad(sizeof(int32_t) * Tx) + kwy_pad(sizeof(uint32_t) * Tx) + kwy_pad(sizeof(uint64_t) * (Tx + 1));
  tot += kwy_pad(sizeof(double) * 2 * (Ty + 2));
  tot += 3 * kwy_pad(sizeof(int32_t) * 2 * (Tx + Ty + 2)) + 2 * kwy_pad(64) + kwy_pad(64);
  return tot + 16 * 256;
}

static int fastdtw_core(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
                        int radius, bool full, double *d_dist, int32_t *d_path, int64_t *d_path_len,
                        int **status_out) {
  std::vector<dtw_level> lv;
  std::vector<const double *> xs, ys;
  lv.push_back({(int)Tx, (int)Ty});
  xs.push_back(x); ys.push_back(y);
  while (!(lv.back().len_x < radius + 2 || lv.back().len_y < radius + 2)) {
    dtw_level c = {lv.back().len_x / 2, lv.back().len_y / 2};
    double *cx = kwy_arena<double>(ctx, (size_t)c.len_x * dim);
    double *cy = kwy_arena<double>(ctx, (size_t)c.len_y * dim);
    if (!cx || !cy) { ctx->err = "fastdtw: scratch arena too small"; return KWY_ENOMEM; }
    hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)c.len_x * dim + 255) / 256)), dim3(256), 0,
                       ctx->stream, xs.back(), c.len_x, dim, cx);
    hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)c.len_y * dim + 255) / 256)), dim3(256), 0,
                       ctx->stream, ys.back(), c.len_y, dim, cy);
    lv.push_back(c); xs.push_back(cx); ys.push_back(cy);
  }
  const uint64_t cap = dtw_cap((int)Tx, (int)Ty, radius, full);
  double *dist = kwy_arena<double>(ctx, cap + 2 * DTW_PAD);
  unsigned long long *pred = kwy_arena<unsigned long long>(ctx, cap / 32 + Tx + 64);
  int32_t *lo = kwy_arena<int32_t>(ctx, Tx), *hi = kwy_arena<int32_t>(ctx, Tx);
  uint32_t *width = kwy_arena<uint32_t>(ctx, Tx);
  uint64_t *off = kwy_arena<uint64_t>(ctx, Tx + 1);
  double *bnd = kwy_arena<double>(ctx, 2 * (Ty + 2));
  int32_t *pathA = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
  int32_t *pathB = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
  int32_t *rev = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
  int64_t *lenA = kwy_arena<int64_t>(ctx, 8), *lenB = kwy_arena<int64_t>(ctx, 8);
  int *status = kwy_arena<int>(ctx, 16);
  if (!dist || !pred || !lo || !hi || !width || !off || !bnd || !pathA || !pathB || !rev || !lenA || !lenB || !status) {
    ctx->err = "fastdtw: scratch arena too small";
    return KWY_ENOMEM;
  }
  *status_out = status;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int) * 16, ctx->stream));
  const size_t bnd_bytes = sizeof(double) * 2 * (Ty + 2);
  const bool bnd_lds = DTW_BT_BYTES + bnd_bytes <= 150 * 1024;
  const size_t dp_lds = DTW_BT_BYTES + (bnd_lds ? bnd_bytes : 0);
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_dp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds));

  const int32_t *cpath = nullptr;
  const int64_t *clen = nullptr;
  for (int l = (int)lv.size() - 1; l >= 0; --l) {
    const int len_x = lv[l].len_x, len_y = lv[l].len_y;
    const bool top = (l == 0);
    int32_t *opath = top ? d_path : ((l & 1) ? pathA : pathB);
    int64_t *olen = top ? d_path_len : ((l & 1) ? lenA : lenB);
    hipLaunchKernelGGL(k_dtw_window, dim3((len_x + 255) / 256), dim3(256), 0, ctx->stream, cpath, clen, radius,
                       len_x, len_y, lo, hi, width);
    KWY_TRY(kwy_launch_scan(ctx, width, off, len_x));
    KWY_PROF(ctx, "k_dtw_dist", hipLaunchKernelGGL(k_dtw_dist, dim3(len_x), dim3(KWY_THREADS), sizeof(double) * dim, ctx->stream, xs[l],
                       ys[l], dim, lo, hi, off, cap, dist, status));
    if (bnd_lds)
      KWY_PROF(ctx, "k_dtw_dp", hipLaunchKernelGGL(k_dtw_dp<true>, dim3(1), dim3(64), dp_lds, ctx->stream, len_x, len_y,
                                                     lo, hi, off, cap, dist, pred, (double *)nullptr, opath, rev, olen,
                                                     d_dist, status));
    else
      KWY_PROF(ctx, "k_dtw_dp", hipLaunchKernelGGL(k_dtw_dp<false>, dim3(1), dim3(64), dp_lds, ctx->stream, len_x, len_y,
                                                     lo, hi, off, cap, dist, pred, bnd, opath, rev, olen, d_dist,
                                                     status));
    cpath = opath;
    clen = olen;
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int dtw_check(kwy_ctx *ctx, const void *x, int64_t Tx, const void *y, int64_t Ty, int dim, int radius,
                     const void *dist, const void *path, const void *path_len) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !y || !dist || !path || !path_len || Tx <= 0 || Ty <= 0 || dim <= 0 || dim > 4096 || radius < 0 ||
      Tx > 0x3fffffff || Ty > 0x3fffffff) {
    ctx->err = "fastdtw: bad argument";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_fastdtw_dev(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
                               int radius, double *dist, int32_t *path, int64_t *path_len) {
  KWY_TRY(dtw_check(ctx, x, Tx, y, Ty, dim, radius, dist, path, path_len));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, dtw_scratch_bytes(Tx, Ty, dim, radius, false)));
  int *status;
  return fastdtw_core(ctx, x, Tx, y, Ty, dim, radius, false, dist, path, path_len, &status);
}

extern "C" int kwy_fastdtw(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
                           int radius, double *dist, int32_t *path, int64_t *path_len) {
  KWY_TRY(dtw_check(ctx, x, Tx, y, Ty, dim, radius, dist, path, path_len));
  KWY_HIP(hipSetDevice(ctx->device));
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool full = attempt == 1;
    size_t bx = kwy_pad(sizeof(double) * Tx * dim), by = kwy_pad(sizeof(double) * Ty * dim);
    size_t bp = kwy_pad(sizeof(int32_t) * 2 * (Tx + Ty + 2));
    KWY_TRY(kwy_arena_begin(ctx, dtw_scratch_bytes(Tx, Ty, dim, radius, full) + bx + by + bp + 2 * kwy_pad(64)));
    double *dx = kwy_arena<double>(ctx, (size_t)Tx * dim), *dy = kwy_arena<double>(ctx, (size_t)Ty * dim);
    int32_t *dpath = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
    double *ddist = kwy_arena<double>(ctx, 8);
    int64_t *dlen = kwy_arena<int64_t>(ctx, 8);
    KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * Tx * dim, hipMemcpyHostToDevice, ctx->stream));
    KWY_HIP(hipMemcpyAsync(dy, y, sizeof(double) * Ty * dim, hipMemcpyHostToDevice, ctx->stream));
    int *status;
    KWY_TRY(fastdtw_core(ctx, dx, Tx, dy, Ty, dim, radius, full, ddist, dpath, dlen, &status));
    int hstatus = 0;
    int64_t hlen = 0;
    KWY_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipMemcpyAsync(&hlen, dlen, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipMemcpyAsync(dist, ddist, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipStreamSynchronize(ctx->stream));
    if (hstatus != 0) continue;  // band storage estimate exceeded: retry with the full matrix
    if (hlen > Tx + Ty) { ctx->err = "fastdtw: path overflow"; return KWY_EHIP; }
    KWY_HIP(hipMemcpy(path, dpath, sizeof(int32_t) * 2 * hlen, hipMemcpyDeviceToHost));
    *path_len = hlen;
    return KWY_OK;
  }
  ctx->err = "fastdtw: band storage overflow";
  return KWY_EHIP;
}
