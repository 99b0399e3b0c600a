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
//   k_dtw_dp       one workgroup: rows are processed in strips of 64, lane = row,
//                  skewed so that lane l works on column j - l; the three
//                  predecessors arrive by wave shuffles (DPP), the strip boundary
//                  row goes through LDS, four wavefronts pipeline the strips.  Every
//                  cell also carries the column at which its best path entered the
//                  strip, so that the back-trace first hops from strip to strip and
//                  then walks all strips at once, one lane per strip.
//
// Tie-breaking follows fastdtw's pure-Python min(): (i-1,j), (i,j-1), (i-1,j-1).
#include <math.h>

#include <vector>

#include "kwy_internal.hpp"

#define DTW_PAD 128          // slack cells before/after the band storage

__global__ void k_dtw_halve(const double *__restrict__ in, int n_out, int dim, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n_out * dim) return;
  const int i = (int)(e / dim), k = (int)(e % dim);
  out[e] = (in[(int64_t)(2 * i) * dim + k] + in[(int64_t)(2 * i + 1) * dim + k]) / 2;
}

// All coarsening levels of both series in ONE launch.  A workgroup owns 2^levels consecutive frames of one series
// and halves them level by level in LDS (ping-pong), storing every level: element i of level l is the pairwise
// average tree over the original frames [i 2^l, (i+1) 2^l) -- the same operations in the same order as halving
// level after level, so the values are bit-identical to k_dtw_halve's.  (n_l = n >> l: the odd frame left over at
// a level is dropped there, as fastdtw's x[:len(x) // 2 * 2] does.)
#define DTW_MAXLV 12
struct dtw_halve_desc {
  const double *src[2];
  double *lv[2][DTW_MAXLV];     // lv[s][l - 1]: level l of series s
  int n[2];
  int chunks0;                  // workgroups of series 0
  int levels, dim;
};
__global__ __launch_bounds__(KWY_THREADS) void k_dtw_halve_all(dtw_halve_desc d) {
  extern __shared__ double hb[];
  const int s = blockIdx.x < d.chunks0 ? 0 : 1;
  const int c = s == 0 ? blockIdx.x : blockIdx.x - d.chunks0;
  const int C = 1 << d.levels, dim = d.dim, n = d.n[s];
  double *A = hb, *B = hb + (size_t)C * dim;
  const int64_t f0 = (int64_t)c * C;
  const int have = (int)min((int64_t)C, (int64_t)n - f0);
  for (int e = threadIdx.x; e < have * dim; e += KWY_THREADS) A[e] = d.src[s][f0 * dim + e];
  __syncthreads();
  for (int l = 1; l <= d.levels; ++l) {
    const int nl = n >> l, per = C >> l;
    const int i0 = c * per;
    const int cnt = min(per, nl - i0);           // level-l elements of this workgroup (may be <= 0)
    double *out = d.lv[s][l - 1];
    for (int e = threadIdx.x; e < cnt * dim; e += KWY_THREADS) {
      const int i = e / dim, k = e - i * dim;
      const double v = (A[(2 * i) * dim + k] + A[(2 * i + 1) * dim + k]) / 2;
      B[e] = v;
      out[(int64_t)(i0 + i) * dim + k] = v;
    }
    __syncthreads();
    double *t = A; A = B; B = t;
  }
}

// Window rows.  cpath == nullptr: full window (the coarsest level).  Otherwise cpath is the coarser level's path (cn
// cells, both coordinates non-decreasing): row i takes the columns of the path cells within +-radius rows of i / 2,
// widened by the radius and doubled.
// The same windows for all rows, and the rows' offsets in the band storage (exclusive prefix sums of the widths), in
// one single-workgroup launch.
// cpath: the coarser level's path (cn cells), or null for the full window of the coarsest level.  tot: NT uint64 of LDS.
template <int NT>
__device__ __forceinline__ void dtw_window_scan_body(const int32_t *__restrict__ cpath, int cn, int radius, int len_x,
                                                     int len_y, int32_t *__restrict__ lo, int32_t *__restrict__ hi,
                                                     uint64_t *__restrict__ off, uint64_t *tot) {
  for (int i = threadIdx.x; i < len_x; i += NT) {
    int l = 0, h = len_y - 1;
    if (cpath && cn > 0) {
      const int a = i / 2;
      int b0 = 0, b1 = cn;
      while (b0 < b1) { int mid = (b0 + b1) >> 1; if (cpath[2 * mid] >= a - radius) b1 = mid; else b0 = mid + 1; }
      const int first = min(b0, cn - 1);
      b0 = 0; b1 = cn;
      while (b0 < b1) { int mid = (b0 + b1) >> 1; if (cpath[2 * mid] > a + radius) b1 = mid; else b0 = mid + 1; }
      const int last = max(b0 - 1, 0);
      l = max(0, 2 * (cpath[2 * first + 1] - radius));
      h = min(len_y - 1, 2 * (cpath[2 * last + 1] + radius) + 1);
    }
    lo[i] = l;
    hi[i] = h;
  }
  __syncthreads();
  kwy_block_count_scan<NT>([&](int64_t i) -> uint64_t { return (uint64_t)(hi[i] - lo[i] + 1); }, len_x, off, tot);
}

// The coarsest level (full window) has its own launch, which also clears the status words of the call; every other
// level's windows are computed by the tail of the previous level's k_dtw_dp (same workgroup, path still hot).
#define DTW_WS_NT 1024
__global__ __launch_bounds__(DTW_WS_NT) void k_dtw_window_scan(const int32_t *__restrict__ cpath,
                                                              const int64_t *__restrict__ cpath_len, int radius,
                                                              int len_x, int len_y, int32_t *__restrict__ lo,
                                                              int32_t *__restrict__ hi, uint64_t *__restrict__ off,
                                                              int *__restrict__ status_clear) {
  __shared__ uint64_t tot[DTW_WS_NT];
  if (status_clear && threadIdx.x < 16) status_clear[threadIdx.x] = 0;
  const int cn = cpath ? (int)*cpath_len : 0;
  dtw_window_scan_body<DTW_WS_NT>(cpath, cn, radius, len_x, len_y, lo, hi, off, tot);
}

// Band storage: row i holds width[i] distances at dist[DTW_PAD + off[i] + 16 i + 8 ...], with DTW_ROWPAD +inf
// cells before and after them.  The DP fetches 8 consecutive cells per lane at a time from a clamped
// start: a lane that is outside its row (wholly or partly) reads pad cells, so a cell outside the window
// costs +inf by itself and the DP step needs no activity mask.
#define DTW_ROWPAD 8
__device__ __forceinline__ uint64_t dtw_row_base(const uint64_t *__restrict__ off, int i) {
  return (uint64_t)DTW_PAD + off[i] + (uint64_t)(2 * DTW_ROWPAD) * (uint64_t)i + DTW_ROWPAD;
}

// dist[row_base(i) + j - lo[i]] = || x_i - y_j ||_2   (sequential sum over the dimensions)
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
  double *row = dist + dtw_row_base(off, i);
  const int l = lo[i], h = hi[i];
  if (threadIdx.x < DTW_ROWPAD) { row[-1 - (int)threadIdx.x] = INFINITY; row[h - l + 1 + threadIdx.x] = INFINITY; }
  for (int j = l + threadIdx.x; j <= h; j += KWY_THREADS) {
    const double *yr = y + (int64_t)j * dim;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) { double d = xs[k] - yr[k]; s += d * d; }
    row[j - l] = sqrt(s);
  }
}

// lane l <- lane l-1 across the whole wavefront (DPP wave_shr:1, no LDS round trip);
// lane 0, which has no source, gets its own lane of `first`.
__device__ __forceinline__ double dtw_wave_shr1(double v, double first) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int dtw_wave_shr1_i32(int v, int first) {
  return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}
// lane l <- lane l+1, lane 63 <- lane 0 (DPP wave_rol:1: every lane has a source)
__device__ __forceinline__ double dtw_wave_rol1(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Predecessor codes: 2 bits per cell, 16 cells per 32-bit word.
// Words are cut at multiples of 16 of the STEP index s (cell (i, j) of strip i0 is handled at
// step s = j - lo[i0] + (i - i0)), so that all lanes of a wavefront flush their word at the same
// step: one store instruction per 16 steps instead of a trickle that the in-order memory counter
// would make every distance prefetch wait for.  A row therefore owns up to width/16 + 2 words.
__device__ __forceinline__ uint64_t dtw_word_base(const uint64_t *__restrict__ off, int i) {
  return (off[i] >> 4) + 2ull * (uint64_t)i;
}
__device__ __forceinline__ int dtw_row_shift(const int32_t *__restrict__ lo, int i) {
  const int i0 = i & ~63;
  return (i - i0) + lo[i] - lo[i0];  // step at which the row's first cell is handled
}
__device__ __forceinline__ uint64_t dtw_row_words_end(const uint64_t *__restrict__ off,
                                                      const int32_t *__restrict__ lo,
                                                      const int32_t *__restrict__ hi, int i) {
  const int sh = dtw_row_shift(lo, i);
  return dtw_word_base(off, i) + (uint64_t)(((sh + hi[i] - lo[i]) >> 4) - (sh >> 4)) + 1;
}

// The DP + back-trace of one level: ONE workgroup of DTW_WAVES wavefronts.  Strip k (64 rows) is
// processed by wave k % DTW_WAVES; strip k+1 trails strip k by one 64-step block, synchronised
// through a progress word in LDS (the boundary row written by strip k is read by strip k+1).
#define DTW_WAVES 4
// Steps between two looks at the previous strip's progress.  Strip k+1 trails strip k by the 63
// steps of the row skew plus one chunk; with 4 wavefronts a wavefront's next strip is ready when it
// finishes the current one only if 4 x (63 + chunk) stays below the strip length (~300 steps).
#define DTW_CHUNK 16
#define DTW_RING 4           // chunks of distances held in registers (three of them in flight)
typedef double dtw_d2 __attribute__((ext_vector_type(2), aligned(8)));
typedef int dtw_i4 __attribute__((ext_vector_type(4), aligned(4)));
template <bool BND_LDS>
__global__ __launch_bounds__(64 * DTW_WAVES) void k_dtw_dp(int len_x, int len_y, const int32_t *lo, const int32_t *hi,
                                              const uint64_t *off /* = lo_w, hi_w, off_w: rewritten by the tail */,
                                              uint64_t cap,
                                              double *dist /* the strips' last rows get their cells' entry columns */,
                                              uint32_t *__restrict__ predw,
                                              double *__restrict__ bnd_global /* DTW_WAVES x (len_y+2) or null */,
                                              int32_t *__restrict__ path, int32_t *__restrict__ rev,
                                              int32_t *__restrict__ sinfo /* 3 x strips */,
                                              int64_t *__restrict__ path_len, double *__restrict__ out_dist,
                                              const int *__restrict__ status, long long *__restrict__ dbg,
                                              int lds_bytes, int next_len_x, int next_len_y, int radius,
                                              int32_t *lo_w, int32_t *hi_w, uint64_t *off_w) {
  extern __shared__ unsigned char bt[];  // boundary rows (BND_LDS); the back-trace's tables afterwards
  const long long t_start = dbg ? clock64() : 0;
  __shared__ int s_n;
  // (strip << 32) | (last finished column + 1).  Plain LDS words written/read with relaxed
  // workgroup-scope atomics: a volatile (generic) access would be a flat_ instruction with a
  // vmcnt(0) wait behind it, i.e. every progress update would also wait for the distance
  // prefetches in flight.  Ordering against the boundary-row accesses needs no wait either: the
  // LDS executes one wavefront's instructions in issue order.
  __shared__ long long s_prog[DTW_WAVES];
#define DTW_PROG_LOAD(b) __hip_atomic_load(&s_prog[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define DTW_PROG_STORE(b, v) __hip_atomic_store(&s_prog[b], (long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// (one workgroup = one CU: its wavefronts share the L1, so global data needs workgroup scope only; a device-scope
// fence writes the L2 back and costs ~20 us each)
#define DTW_RELEASE() do { if (BND_LDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); } while (0)
#define DTW_ACQUIRE() do { if (BND_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (*status != 0) {
    if (threadIdx.x == 0) { *path_len = 0; *out_dist = NAN; }
    // keep the next level's tables defined (full windows: its distance kernel sees the overflow and leaves the status)
    if (next_len_x > 0)
      dtw_window_scan_body<64 * DTW_WAVES>((const int32_t *)nullptr, 0, radius, next_len_x, next_len_y, lo_w, hi_w, off_w,
                                           (uint64_t *)bt);
    return;
  }
  const double INF = INFINITY;
  const int rowlen = len_y + 2;  // boundary rows are indexed by j + 1 (entry 0 is column -1)
  double *const lds_rows = (double *)bt;
#define BROW(buf, ix) (BND_LDS ? lds_rows[(buf) * rowlen + (ix)] : bnd_global[(size_t)(buf) * rowlen + (ix)])
  if (threadIdx.x < DTW_WAVES) DTW_PROG_STORE(threadIdx.x, -1);
  __syncthreads();
  const int nstrips = (len_x + 63) / 64;
  for (int k = wv; k < nstrips; k += DTW_WAVES) {
    const int i0 = k * 64;
    const int i = i0 + lane;
    const bool valid = i < len_x;
    const int rl = valid ? lo[i] : 0, rh = valid ? hi[i] : -1 - DTW_ROWPAD;
    const int rw = rh - rl;  // last valid index of the row (no row: every fetch lands in the left pad)
    const int ilast = min(i0 + 63, len_x - 1);
    const int L = ilast - i0;  // lane of the strip's last row
    const int jmin = lo[i0], jmax = hi[ilast];
    // the previous strip's last row: where its boundary values are valid
    const int plo = k > 0 ? lo[i0 - 1] : 0, phi = k > 0 ? hi[i0 - 1] : -1;
    const int pbuf = (k + DTW_WAVES - 1) % DTW_WAVES, nbuf = k % DTW_WAVES;
    // distances: a scalar base per strip (the first row's left pad) plus a 32-bit per-lane byte offset
    char *dbase = (char *)(dist + dtw_row_base(off, i0) - DTW_ROWPAD);
    const uint32_t boff = (uint32_t)((dtw_row_base(off, valid ? i : i0) - dtw_row_base(off, i0)) * 8ull);
    uint32_t *pwrow = predw + (valid ? dtw_word_base(off, i) : 0);
    double v1 = INF;      // this lane's value at the previous step
    double up_prev = INF; // the `up` input of the previous step = this step's diagonal input
    // entry column of the best path into this strip, per cell (same selection as the value): lane 0's
    // `up` / diagonal cells are boundary cells, which are their own entry columns
    int o1 = 0, oup_prev = jmin - 1;
    const int nsteps = (jmax - jmin + 1) + 63;
    if (dbg && lane == 0 && k < 48) { dbg[64 + 4 * k] = clock64() - t_start; dbg[64 + 4 * k + 2] = nsteps; dbg[64 + 4 * k + 3] = jmin; }
    const int shift = lane + rl - jmin;  // this lane's row index at step s is s - shift
    // the last row's range of steps and last column, as scalars: its boundary writes are guarded on the scalar unit
    const int shL = __builtin_amdgcn_readlane(shift, L), rwL = __builtin_amdgcn_readlane(rw, L);
    const int rhL = __builtin_amdgcn_readlane(rh, L);
    uint32_t pw = 0u;                    // predecessor codes of the current 16-cell word
    // lane 0's diagonal input at the first step: D[i0-1][jmin-1]
    if (k == 0 && lane == 0 && jmin == 0) up_prev = 0.0;  // D[-1][-1] = 0: the origin of the recurrence
    // Distances of this lane's row: a ring of DTW_RING chunks of 16 steps in registers, each fetched
    // DTW_RING - 1 chunks (48 steps) before it is used -- the band was written by another kernel on
    // other XCDs and comes from memory (about 3500 cycles on an otherwise idle chip), and a step is
    // only ~100 cycles.  Two 8-cell fetches per chunk, each from a clamped start (see DTW_ROWPAD).
    double ring[DTW_RING][DTW_CHUNK];
    auto fetch = [&](double (&b)[DTW_CHUNK], int cstart) {
#pragma unroll
      for (int half = 0; half < DTW_CHUNK / 8; ++half) {
        const int start = min(max(cstart + 8 * half - shift, -DTW_ROWPAD), rw + 1);
        const dtw_d2 *p = (const dtw_d2 *)(dbase + (boff + 8u * (uint32_t)(start + DTW_ROWPAD)));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const dtw_d2 v = p[q];
          b[8 * half + 2 * q] = v.x;
          b[8 * half + 2 * q + 1] = v.y;
        }
      }
    };
#pragma unroll
    for (int r = 0; r < DTW_RING - 1; ++r) fetch(ring[r], DTW_CHUNK * r);
    bool first_chunk = true;
    // one chunk of DTW_CHUNK steps on `cur`; `fill` (the buffer used one chunk ago) is refilled meanwhile
    auto chunk = [&](int c0, const double (&cur)[DTW_CHUNK], double (&fill)[DTW_CHUNK]) {
      fetch(fill, c0 + DTW_CHUNK * (DTW_RING - 1));
      // wait until the previous strip's last row has produced the columns this chunk reads
      if (k > 0) {
        const int need = min(jmin + c0 + DTW_CHUNK - 1, phi);  // last column we may read (valid ones only)
        const long long tw0 = dbg ? clock64() : 0;
        while (true) {
          const long long pv = DTW_PROG_LOAD(pbuf);
          const int ps = (int)(pv >> 32), pc = (int)(pv & 0xffffffffll) - 1;
          if (ps > k - 1 || (ps == k - 1 && pc >= need)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        DTW_ACQUIRE();
        if (dbg && lane == 0) { atomicAdd((unsigned long long *)&dbg[8 + wv], (unsigned long long)(clock64() - tw0)); }
      }
      if (dbg && lane == 0) atomicAdd((unsigned long long *)&dbg[12 + wv], 1ull);
      // the boundary values lane 0 needs in the next DTW_CHUNK steps: lane t holds the one of step c0 + t;
      // the vector is rotated by one lane per step, so that lane 0 always holds the current one
      const int jb = jmin + c0 + lane;
      double brot = INF;
      if (k > 0 && jb >= plo && jb <= phi) brot = BROW(pbuf, jb + 1);
      if (first_chunk && k > 0) {
        const int jd = jmin - 1;
        const double dv = (jd >= plo && jd <= phi) ? BROW(pbuf, jd + 1) : INF;
        if (lane == 0) up_prev = dv;
        first_chunk = false;
      }
#pragma unroll
      for (int half = 0; half < DTW_CHUNK / 8; ++half) {
        const int s0 = c0 + 8 * half;
        double hist[8];
        int ohist[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          // up = D[i-1][j]: the neighbouring lane's value of the previous step (lane 0: the boundary row).
          // The diagonal D[i-1][j-1] is what `up` was one step ago -- no second shift.
          const double nbrot = dtw_wave_rol1(brot);
          const double up = dtw_wave_shr1(v1, brot);   // brot dies here: the shift lands in its register
          brot = nbrot;
          const double dg = up_prev;
          up_prev = up;
          // a lane outside its row adds +inf: every candidate is +inf, the code stays 0
          const double dt = cur[8 * half + u];
          // fastdtw's min() keeps the first of equal candidates in the order (i-1,j), (i,j-1), (i-1,j-1):
          // the value is the plain minimum (two v_min_f64 on the loop-carried chain instead of two
          // compare-and-select pairs), the code is the first candidate equal to it (off the chain)
          const double c0v = up + dt, c1v = v1 + dt, c2v = dg + dt;
          const double best = fmin(c0v, fmin(c1v, c2v));
          const bool e0 = best == c0v, e1 = best == c1v;
          const uint32_t pb = e0 ? 0u : (e1 ? 1u : 2u);
          pw |= pb << (2 * (8 * half + u));   // c0 is a multiple of 16: the step's place in its word
          const int oup = dtw_wave_shr1_i32(o1, jmin + s0 + u);
          const int odg = oup_prev;
          oup_prev = oup;
          o1 = e0 ? oup : (e1 ? o1 : odg);
          ohist[u] = o1;
          hist[u] = best;
          v1 = best;
          // keep the compare masks of at most four steps alive (left alone, the scheduler collects the
          // masks of all 16 steps in SGPRs and spills them)
          if (u == 3 || u == 7) __builtin_amdgcn_sched_barrier(0);
        }
        // the strip's last row goes to the boundary buffer (its lane only, the steps inside its row only)
        if (s0 >= shL && s0 + 7 <= shL + rwL) {          // wave-uniform: the whole block lies inside the row
          if (lane == L) {
#pragma unroll
            for (int u = 0; u < 8; ++u) BROW(nbuf, jmin + s0 - L + 1 + u) = hist[u];
            // the entry columns, as an int array laid over the start of the row's own distances (entry p
            // sits inside cell p/2, which this lane fetched long ago)
            dtw_i4 *op = (dtw_i4 *)(dbase + (boff + 8u * DTW_ROWPAD + 4u * (uint32_t)(s0 - shift)));
            op[0] = dtw_i4{ohist[0], ohist[1], ohist[2], ohist[3]};
            op[1] = dtw_i4{ohist[4], ohist[5], ohist[6], ohist[7]};
          }
        } else if (s0 + 7 >= shL && s0 <= shL + rwL) {   // the block straddles one of the row's ends
          if (lane == L) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int s = s0 + u;
              if (s >= shL && s <= shL + rwL) {
                BROW(nbuf, jmin + s - L + 1) = hist[u];
                *(int32_t *)(dbase + (boff + 8u * DTW_ROWPAD + 4u * (uint32_t)(s - shift))) = ohist[u];
              }
            }
          }
        }
        // publish how far this strip's last row has got (after its boundary writes)
        DTW_RELEASE();
        if (lane == L) {
          const int jdone = min(jmin + (s0 + 7) - L, rhL);
          DTW_PROG_STORE(nbuf, ((long long)k << 32) | (long long)(unsigned int)(jdone + 1 > 0 ? jdone + 1 : 0));
        }
      }
      // one predecessor word per lane and chunk, if the lane's row has cells in it
      if (valid && shift <= c0 + DTW_CHUNK - 1 && shift + rw >= c0) pwrow[(c0 >> 4) - (shift >> 4)] = pw;
      pw = 0u;
    };
    // whole chunks (the steps behind nsteps see +inf only), the ring's phases unrolled
    for (int c0 = 0; c0 < nsteps; c0 += DTW_RING * DTW_CHUNK) {
      chunk(c0, ring[0], ring[3]);
      if (c0 + DTW_CHUNK >= nsteps) break;
      chunk(c0 + DTW_CHUNK, ring[1], ring[0]);
      if (c0 + 2 * DTW_CHUNK >= nsteps) break;
      chunk(c0 + 2 * DTW_CHUNK, ring[2], ring[1]);
      if (c0 + 3 * DTW_CHUNK >= nsteps) break;
      chunk(c0 + 3 * DTW_CHUNK, ring[3], ring[2]);
    }
    DTW_RELEASE();
    if (lane == L) DTW_PROG_STORE(nbuf, ((long long)k << 32) | 0x7fffffffll);
    if (dbg && lane == 0 && k < 48) dbg[64 + 4 * k + 1] = clock64() - t_start;
  }
  __syncthreads();
  // D[len_x-1][len_y-1]: the last strip's last row is in its boundary buffer
  const double last_val = BROW((nstrips - 1) % DTW_WAVES, len_y);
#undef BROW
  __syncthreads();   // every thread holds last_val: the back-trace staging may overwrite the boundary rows
  const long long t_dp = dbg ? clock64() : 0;
  if (threadIdx.x == 0) *out_dist = last_val;
  __syncthreads();   // predecessor codes and entry columns written above are read back below through global memory
                     // (by this workgroup only: the barrier's workgroup-scope fence is enough)

  // ---- back-trace.  (1) one thread hops from strip to strip: the path leaves strip k through
  //      (last row, exitc[k]) and the entry column stored there is where it leaves strip k-1.
  //      (2) every strip is walked by its own lane, all at once, over the predecessor codes.
  //      (3) the strips' cell counts are summed, (4) the segments are copied to their places.
  // A walk is a chain of dependent loads (row window -> word address -> predecessor word), ~1500 cycles per cell
  // from global memory: the per-row table {lo, hi, first word} and, when they fit, the predecessor words are
  // staged in LDS first (the boundary rows are dead by now).
  int32_t *exitc = sinfo, *sbase = sinfo + nstrips, *cnt = sinfo + 2 * nstrips;
  const bool tbl_ok = (size_t)lds_bytes >= 12ull * (size_t)len_x;
  int32_t *t_lo = (int32_t *)bt, *t_hi = t_lo + len_x;
  uint32_t *t_wb = (uint32_t *)(t_hi + len_x), *t_pw = t_wb + len_x;
  const uint64_t nwords = dtw_row_words_end(off, lo, hi, len_x - 1);
  const bool pw_ok = tbl_ok && nwords <= (uint64_t)(((size_t)lds_bytes - 12ull * (size_t)len_x) / 4);
  if (tbl_ok) {
    for (int i = threadIdx.x; i < len_x; i += 64 * DTW_WAVES) {
      t_lo[i] = lo[i]; t_hi[i] = hi[i]; t_wb[i] = (uint32_t)dtw_word_base(off, i);
    }
    if (pw_ok) for (uint32_t w = threadIdx.x; w < (uint32_t)nwords; w += 64 * DTW_WAVES) t_pw[w] = predw[w];
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[20] = clock64() - t_dp;
  // Two instances of hop and walk: on the LDS tables (ds_ loads: they do not share a wait counter with the global
  // stores of the walk's output) or on the arrays in global memory.  A flat pointer serving both would turn every
  // load into a flat_ instruction that waits for all outstanding stores.
  auto hop = [&](const int32_t *LO, const int32_t *HI) {
    int cj = len_y - 1, base = 0;
    for (int k = nstrips - 1; k >= 0; --k) {
      const int i0 = 64 * k, il = min(i0 + 63, len_x - 1);
      exitc[k] = cj;
      int e = 0;
      if (k > 0) {
        const int l = LO[il];
        const int c = min(max(cj, l), HI[il]);
        e = ((const int32_t *)(dist + dtw_row_base(off, il)))[c - l];
        e = min(max(e, 0), cj);
      }
      sbase[k] = base;
      base += (il - i0 + 1) + (cj - e) + 1;   // rows + columns: more cells than the strip can hold
      cj = e;
    }
  };
  if (threadIdx.x == 0) {
    if (tbl_ok) hop(t_lo, t_hi); else hop(lo, hi);
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[21] = clock64() - t_dp;
  auto walk = [&](int k, const int32_t *LO, const int32_t *HI, const uint32_t *WB, const uint32_t *PW) {
    const int i0 = 64 * k, lo0 = LO[i0];
    int i = min(i0 + 63, len_x - 1), j = exitc[k], m = 0;
    int32_t *out = rev + 2 * (int64_t)sbase[k];
    int crow = -1, l = 0, h = -1, sh = 0;
    uint64_t wb = 0, cw = ~0ull;
    uint32_t word = 0u;
    while (i >= i0) {
      out[2 * m] = i; out[2 * m + 1] = j; ++m;
      if (i == 0 && j == 0) break;
      if (i != crow) {
        crow = i; l = LO[i]; h = HI[i]; sh = (i - i0) + l - lo0;
        wb = WB ? (uint64_t)WB[i] : dtw_word_base(off, i);
        cw = ~0ull;
      }
      unsigned int pb = 0;
      if (j >= l && j <= h) {
        const int st = sh + (j - l);  // the step at which this cell was computed
        const uint64_t w = wb + (uint64_t)((st >> 4) - (sh >> 4));
        if (w != cw) { cw = w; word = PW[w]; }
        pb = (word >> (2 * (st & 15))) & 3u;
      }
      if (pb == 0) --i; else if (pb == 1) --j; else { --i; --j; }
      if (j < 0) break;
    }
    cnt[k] = m;
  };
  for (int k = threadIdx.x; k < nstrips; k += 64 * DTW_WAVES) {
    if (pw_ok) walk(k, t_lo, t_hi, t_wb, t_pw);
    else if (tbl_ok) walk(k, t_lo, t_hi, t_wb, predw);
    else walk(k, lo, hi, (const uint32_t *)nullptr, predw);
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[22] = clock64() - t_dp;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int k = 0; k < nstrips; ++k) { const int c = cnt[k]; exitc[k] = n; n += c; }   // exitc: now the output offset
    s_n = n;
    *path_len = n;
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[23] = clock64() - t_dp;
  // every strip's segment, reversed, to its place: one thread per path cell (the strip of an output position by
  // binary search over the segment offsets, staged in LDS) -- one lane per strip used to copy its ~130 cells one
  // dependent load at a time
  {
    constexpr int MAXS = 256;
    __shared__ int s_meta[3][MAXS];
    if (nstrips <= MAXS) {
      for (int k = threadIdx.x; k < nstrips; k += 64 * DTW_WAVES) {
        s_meta[0][k] = cnt[k]; s_meta[1][k] = sbase[k]; s_meta[2][k] = exitc[k];
      }
      __syncthreads();
      const int n = s_n;
      for (int pos = threadIdx.x; pos < n; pos += 64 * DTW_WAVES) {
        int a = 0, b = nstrips - 1;                  // last strip whose offset is <= pos
        while (a < b) { const int mid = (a + b + 1) >> 1; if (s_meta[2][mid] <= pos) a = mid; else b = mid - 1; }
        const int c = s_meta[0][a], q = pos - s_meta[2][a];
        ((int2 *)path)[pos] = ((const int2 *)rev)[(int64_t)s_meta[1][a] + (c - 1 - q)];
      }
    } else {
      for (int k = threadIdx.x; k < nstrips; k += 64 * DTW_WAVES) {
        const int c = cnt[k];
        const int2 *in = (const int2 *)rev + (int64_t)sbase[k];
        int2 *out = (int2 *)path + (int64_t)exitc[k];
        for (int m = 0; m < c; ++m) out[c - 1 - m] = in[m];
      }
    }
  }
  if (dbg && threadIdx.x == 0) {
    const long long t_end = clock64();
    atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)(t_dp - t_start));
    atomicAdd((unsigned long long *)&dbg[1], (unsigned long long)(t_end - t_dp));
    dbg[2] = t_dp - t_start; dbg[3] = t_end - t_dp; dbg[4] = s_n;
  }
  // ---- the next (finer) level's windows from this path: lo / hi / off of this level are dead now
  if (next_len_x > 0) {
    __syncthreads();    // the path is complete (workgroup scope), the LDS tables are free
    dtw_window_scan_body<64 * DTW_WAVES>(path, s_n, radius, next_len_x, next_len_y, lo_w, hi_w, off_w, (uint64_t *)bt);
  }
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
  tot += kwy_pad(sizeof(double) * (cap + 2 * DTW_ROWPAD * Tx + 2 * DTW_PAD)) + kwy_pad(4 * (cap / 16 + 2 * Tx + 64));
  tot += 2 * kwy_pad(sizeof(int32_t) * Tx) + kwy_pad(sizeof(uint64_t) * (Tx + 1));
  tot += kwy_pad(sizeof(double) * 4 * (Ty + 2));
  tot += 2 * kwy_pad(sizeof(int32_t) * 2 * (Tx + Ty + 2)) + kwy_pad(sizeof(int32_t) * 2 * (Tx + Ty + Tx / 64 + 8)) +
         kwy_pad(sizeof(int32_t) * 3 * (Tx / 64 + 2)) + 2 * kwy_pad(64) + kwy_pad(64);
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
    lv.push_back(c); xs.push_back(cx); ys.push_back(cy);
  }
  {
    const int levels = (int)lv.size() - 1;
    const size_t halve_lds = levels > 0 ? sizeof(double) * (((size_t)3 << levels) / 2) * dim : 0;
    if (levels > 0 && levels <= DTW_MAXLV && halve_lds <= 64 * 1024) {
      dtw_halve_desc d;
      d.src[0] = x; d.src[1] = y;
      d.n[0] = (int)Tx; d.n[1] = (int)Ty;
      d.levels = levels; d.dim = dim;
      for (int l = 1; l <= levels; ++l) { d.lv[0][l - 1] = (double *)xs[l]; d.lv[1][l - 1] = (double *)ys[l]; }
      const int C = 1 << levels;
      d.chunks0 = (int)((Tx + C - 1) / C);
      const int chunks1 = (int)((Ty + C - 1) / C);
      hipLaunchKernelGGL(k_dtw_halve_all, dim3(d.chunks0 + chunks1), dim3(KWY_THREADS), halve_lds, ctx->stream, d);
    } else {
      for (int l = 1; l <= levels; ++l) {      // long series / wide features: level by level
        hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)lv[l].len_x * dim + 255) / 256)), dim3(256), 0,
                           ctx->stream, xs[l - 1], lv[l].len_x, dim, (double *)xs[l]);
        hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)lv[l].len_y * dim + 255) / 256)), dim3(256), 0,
                           ctx->stream, ys[l - 1], lv[l].len_y, dim, (double *)ys[l]);
      }
    }
  }
  const uint64_t cap = dtw_cap((int)Tx, (int)Ty, radius, full);
  double *dist = kwy_arena<double>(ctx, cap + 2 * DTW_ROWPAD * Tx + 2 * DTW_PAD);   // + the +inf cells of every row
  uint32_t *pred = kwy_arena<uint32_t>(ctx, cap / 16 + 2 * Tx + 64);
  int32_t *lo = kwy_arena<int32_t>(ctx, Tx), *hi = kwy_arena<int32_t>(ctx, Tx);
  uint64_t *off = kwy_arena<uint64_t>(ctx, Tx + 1);
  double *bnd = kwy_arena<double>(ctx, DTW_WAVES * (Ty + 2));
  int32_t *pathA = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
  int32_t *pathB = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
  int32_t *rev = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + Tx / 64 + 8));   // the strips' segments, with slack per strip
  int32_t *sinfo = kwy_arena<int32_t>(ctx, 3 * (Tx / 64 + 2));
  int64_t *lenA = kwy_arena<int64_t>(ctx, 8), *lenB = kwy_arena<int64_t>(ctx, 8);
  int *status = kwy_arena<int>(ctx, 16);
  if (!dist || !pred || !lo || !hi || !off || !bnd || !pathA || !pathB || !rev || !sinfo || !lenA || !lenB || !status) {
    ctx->err = "fastdtw: scratch arena too small";
    return KWY_ENOMEM;
  }
  *status_out = status;
  const size_t bnd_bytes = sizeof(double) * DTW_WAVES * (Ty + 2);
  // the boundary rows of the strips live in LDS when they fit
  const bool bnd_lds = bnd_bytes <= 150 * 1024;
  // ... and the back-trace stages its row table and predecessor words in the same LDS: ask for all a workgroup may
  // have beside the static variables (the kernel takes what fits)
  const size_t dp_lds = 150 * 1024;
  (void)bnd_bytes;
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_dp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds));

  const int32_t *cpath = nullptr;
  const int64_t *clen = nullptr;
  for (int l = (int)lv.size() - 1; l >= 0; --l) {
    const int len_x = lv[l].len_x, len_y = lv[l].len_y;
    const bool top = (l == 0);
    int32_t *opath = top ? d_path : ((l & 1) ? pathA : pathB);
    int64_t *olen = top ? d_path_len : ((l & 1) ? lenA : lenB);
    if (l == (int)lv.size() - 1)
      hipLaunchKernelGGL(k_dtw_window_scan, dim3(1), dim3(DTW_WS_NT), 0, ctx->stream, (const int32_t *)nullptr,
                         (const int64_t *)nullptr, radius, len_x, len_y, lo, hi, off, status);
    const int nlx = top ? 0 : lv[l - 1].len_x, nly = top ? 0 : lv[l - 1].len_y;
    KWY_PROF(ctx, "k_dtw_dist", hipLaunchKernelGGL(k_dtw_dist, dim3(len_x), dim3(KWY_THREADS), sizeof(double) * dim, ctx->stream, xs[l],
                       ys[l], dim, lo, hi, off, cap, dist, status));
    if (bnd_lds)
      KWY_PROF(ctx, "k_dtw_dp", hipLaunchKernelGGL(k_dtw_dp<true>, dim3(1), dim3(64 * DTW_WAVES), dp_lds, ctx->stream, len_x, len_y,
                                                     lo, hi, off, cap, dist, pred, (double *)nullptr, opath, rev, sinfo, olen,
                                                     d_dist, status, (long long *)ctx->dbg, (int)dp_lds, nlx, nly, radius,
                                                     lo, hi, off));
    else
      KWY_PROF(ctx, "k_dtw_dp", hipLaunchKernelGGL(k_dtw_dp<false>, dim3(1), dim3(64 * DTW_WAVES), dp_lds, ctx->stream, len_x, len_y,
                                                     lo, hi, off, cap, dist, pred, bnd, opath, rev, sinfo, olen, d_dist,
                                                     status, (long long *)ctx->dbg, (int)dp_lds, nlx, nly, radius, lo, hi,
                                                     off));
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
