// kwy_dtw.hip -- FastDTW (multi-resolution, windowed dynamic time warping) on gfx950.
//
// Replaces fastdtw.fastdtw(x, y, radius, dist=2) (reference call site
// kwiiyatta/vocoder/align.py:71; fastdtw 0.3.2).  The level recursion is unrolled
// on the host (sizes depend only on Tx, Ty, radius); everything else stays on
// the device with no host synchronisation:
//
//   k_dtw_halve_all    all coarsening levels of both series: (x[2i] + x[2i+1]) / 2, level after level in LDS
//   k_dtw_window_scan  the coarsest level's (full) windows; every other level's windows come out of the tail of the
//                      level before (k_dtw_trace): a table {first column, last column} per coarse row, two look-ups
//   k_dtw_dist         Euclidean frame distances for every cell of the strips' rectangles (skewed band: see below)
//   k_dtw_values       ONE workgroup, the recurrence and nothing else: rows in strips of 64, lane = row, lane l works
//                      on column j - l; the neighbours' values arrive by DPP shifts, the strip boundary row goes
//                      through LDS, four wavefronts pipeline the strips; D is written to a second skewed band
//   k_dtw_codes        one workgroup per strip: predecessor of every cell (two bit planes) and, for the cells of the
//                      strip's last row, the column at which the best path entered the strip
//   k_dtw_trace        ONE workgroup: hops from strip to strip over the entry columns, walks all strips at once (one
//                      lane each), copies the segments, computes the next level's windows
//   k_dtw_small        a level of a single strip: distances, recurrence, codes and trace in one launch
//
// Tie-breaking follows fastdtw's pure-Python min(): (i-1,j), (i,j-1), (i-1,j-1).
#include <math.h>

#include <algorithm>
#include <vector>

#include "kwy_internal.hpp"


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

// Window rows.  cpath == nullptr: full window (the coarsest level).  Otherwise cpath is the coarser level's path (cn
// cells, both coordinates non-decreasing): row i takes the columns of the path cells within +-radius rows of i / 2,
// widened by the radius and doubled.
// The same windows for all rows, the rows' offsets in the predecessor planes (exclusive prefix sums of the widths) and
// the strips' offsets in the skewed bands, in one single-workgroup launch.
//
// Skewed band (distances, D values): strip k = rows 64 k .. 64 k + 63.  Cell (i, j) of the strip is handled by lane
// i - 64 k at step s = j - lo[64 k] + (i - 64 k) of the recurrence and lives at
//     soff[k] + ((s >> 1) * 64 + lane) * 2 + (s & 1):
// what the 64 lanes of a wavefront need for two consecutive steps is 1 KB of consecutive memory.  (With one row per
// lane in a row-major band every lane of a load touched its own cache line: 64 tag look-ups per instruction, and
// those look-ups, not the arithmetic, set the 170-200 cycles a step took until round 3.)  A strip holds
// steps16 = ceil16(hi[last row] - lo[first row] + 1 + 63) steps; the cells of the rectangle outside the rows' windows
// hold +inf distances.
__device__ __forceinline__ int dtw_strip_steps16(const int32_t *lo, const int32_t *hi, int len_x, int k) {
  const int il = min(64 * k + 63, len_x - 1);
  return ((hi[il] - lo[64 * k] + 1 + 63) + 15) & ~15;
}
__device__ __forceinline__ uint64_t dtw_skew_index(uint64_t sbase, int s, int lane) {
  return sbase + ((uint64_t)(s >> 1) * 64 + (uint64_t)lane) * 2 + (uint64_t)(s & 1);
}
// cpath: the coarser level's path (cn cells), or null for the full window of the coarsest level.
// lds: lds_bytes of LDS, at least NT uint64.  With room for two ints per coarse row the path is first turned into a
// table {column of the row's first cell, column of its last cell} (a monotone path visits every row: a cell is the
// first of its row if the cell before it lies in another one), and a window is two look-ups; otherwise two binary
// searches over the path in memory per row (12 dependent loads each at the finest level).
template <int NT>
__device__ __forceinline__ void dtw_window_scan_body(const int32_t *__restrict__ cpath, int cn, int radius, int len_x,
                                                     int len_y, int32_t *lo, int32_t *hi, uint64_t *off,
                                                     uint64_t *soff, uint64_t cap_rows, uint64_t cap_skew,
                                                     int *status, unsigned char *lds, size_t lds_bytes) {
  uint64_t *tot = (uint64_t *)lds;
  const int crows = (cpath && cn > 0) ? cpath[2 * (cn - 1)] + 1 : 0;       // rows of the coarser level
  const bool table = crows > 0 && sizeof(uint64_t) * NT + 8ull * (size_t)crows <= lds_bytes;
  int32_t *firstj = (int32_t *)(tot + NT), *lastj = firstj + crows;
  if (table) {
    for (int p = threadIdx.x; p < cn; p += NT) {
      const int r = cpath[2 * p], j = cpath[2 * p + 1];
      if (p == 0 || cpath[2 * (p - 1)] != r) firstj[r] = j;
      if (p == cn - 1 || cpath[2 * (p + 1)] != r) lastj[r] = j;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < len_x; i += NT) {
    int l = 0, h = len_y - 1;
    if (table) {
      const int a = i / 2;
      const int fr = a - radius;             // first cell in a row >= fr (none: the path's last cell)
      const int jf = fr < crows ? firstj[max(fr, 0)] : lastj[crows - 1];
      const int jl = lastj[min(a + radius, crows - 1)];                  // last cell in a row <= a + radius
      l = max(0, 2 * (jf - radius));
      h = min(len_y - 1, 2 * (jl + radius) + 1);
    } else if (cpath && cn > 0) {
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
  const int nstrips = (len_x + 63) / 64;
  kwy_block_count_scan<NT>([&](int64_t k) -> uint64_t { return 64ull * (uint64_t)dtw_strip_steps16(lo, hi, len_x, (int)k); },
                           nstrips, soff, tot);
  __syncthreads();
  if (threadIdx.x == 0 && (off[len_x] > cap_rows || soff[nstrips] > cap_skew)) atomicExch(status, 1);
}

// The coarsest level (full window) has its own launch, which also clears the status words of the call; every other
// level's windows are computed by the tail of the previous level's k_dtw_trace (same workgroup, path still hot).
#define DTW_WS_NT 1024
__device__ __forceinline__ void dtw_window_scan_first(int radius, int len_x, int len_y, int32_t *lo, int32_t *hi,
                                                      uint64_t *off, uint64_t *soff, uint64_t cap_rows,
                                                      uint64_t cap_skew, int *status, uint64_t *tot, size_t tot_bytes) {
  if (threadIdx.x < 16) status[threadIdx.x] = 0;
  __syncthreads();
  dtw_window_scan_body<DTW_WS_NT>((const int32_t *)nullptr, 0, radius, len_x, len_y, lo, hi, off, soff, cap_rows,
                                  cap_skew, status, (unsigned char *)tot, tot_bytes);
}

// dist(i, j) = || x_i - y_j ||_2 (sequential sum over the dimensions) for every cell of the strips' rectangles that
// lies in its row's window, +inf for the others.  A workgroup = 16 steps x 64 lanes of one strip: 64 frames of x and
// the 79 frames of y its cells pair them with, staged in LDS 32 dimensions at a time (the sum keeps its order).
#define DTW_DT 32                       // dimensions per tile
#define DTW_DTP (DTW_DT + 1)            // padded row (lane l reads row l, row s - l: 33 doubles apart, no bank conflict)
#define DTW_DIST_LDS (sizeof(double) * (64 + 80) * DTW_DTP)
// (unit_first, unit_stride: the units of 1024 cells this workgroup takes; lds: DTW_DIST_LDS bytes)
__device__ __forceinline__ void dtw_dist_body(const double *x, const double *y, int dim, int len_x, int len_y,
                                              const int32_t *lo, const int32_t *hi, const uint64_t *soff,
                                              double *dist, const int *status, unsigned unit_first,
                                              unsigned unit_stride, unsigned char *lds) {
  double *xs = (double *)lds, *ys = xs + 64 * DTW_DTP;
  if (status[0] != 0) return;
  const int nstrips = (len_x + 63) / 64;
  const uint64_t total = soff[nstrips];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  constexpr int PER = 16 / (KWY_THREADS / 64);          // steps per wavefront
  for (uint64_t cell0 = (uint64_t)unit_first * 1024ull; cell0 < total; cell0 += (uint64_t)unit_stride * 1024ull) {
    int a = 0, b = nstrips - 1;                   // the strip of this unit: the last one that starts at or before it
    while (a < b) { const int mid = (a + b + 1) >> 1; if (soff[mid] <= cell0) a = mid; else b = mid - 1; }
    const int k = a;
    const uint64_t sb = soff[k];
    const int s0 = (int)((cell0 - sb) >> 6);
    const int i = 64 * k + lane;
    const bool valid = i < len_x;
    const int jmin = lo[64 * k], l = valid ? lo[i] : 0, h = valid ? hi[i] : -1;
    const int jbase = jmin + s0 - 63;             // y frame of ys row 0; the cell (lane, step s) uses row s - s0 + 63 - lane
    bool in[PER];
    double acc[PER];
    bool any = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = jmin + s0 + PER * sub + q - lane;
      in[q] = j >= l && j <= h;
      any = any || in[q];
      acc[q] = 0.0;
    }
    if (__syncthreads_or(any)) {
      for (int c0 = 0; c0 < dim; c0 += DTW_DT) {
        const int cw = min(DTW_DT, dim - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * cw; e += KWY_THREADS) {
          const int r = e / cw, c = e - r * cw, ii = 64 * k + r;
          xs[r * DTW_DTP + c] = ii < len_x ? x[(int64_t)ii * dim + c0 + c] : 0.0;
        }
        for (int e = threadIdx.x; e < 79 * cw; e += KWY_THREADS) {
          const int r = e / cw, c = e - r * cw, jj = jbase + r;
          ys[r * DTW_DTP + c] = (jj >= 0 && jj < len_y) ? y[(int64_t)jj * dim + c0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (in[q]) {
            const double *xr = xs + lane * DTW_DTP, *yr = ys + (PER * sub + q + 63 - lane) * DTW_DTP;
            double t = acc[q];
            for (int c = 0; c < cw; ++c) { const double df = xr[c] - yr[c]; t += df * df; }
            acc[q] = t;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
      dist[dtw_skew_index(sb, s0 + PER * sub + q, lane)] = in[q] ? sqrt(acc[q]) : INFINITY;
    __syncthreads();        // (the tiles are rewritten by the next unit)
  }
}
// (the kernels themselves follow the batch descriptor's helpers: see k_dtw_* below)

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

// The recurrence's minimum.  fmin() puts a canonicalising v_max_f64 x, x in front of an operand that comes out of a
// lane shift (its contract for signalling NaNs): one per step, on the dependence chain.  Measured and left alone: a
// v_min_f64 in inline asm made the compiler pad every chunk with 17 s_nop; compiling the file without NaN
// semantics removes the v_max (517 -> 501 us over the levels of 2201 x 2401) but covers every kernel of the file.
__device__ __forceinline__ double dtw_min_raw(double a, double b) {
  return fmin(a, b);
}

// Predecessor codes: two bit planes per (row, group of 64 columns of the row's window): bit c of m0 says "the first
// candidate (i-1, j) attains the minimum", bit c of m1 the same for (i, j-1); neither: the diagonal.  Group g of row i
// is predm[2 (gbase(i) + g)], predm[2 (gbase(i) + g) + 1].
__device__ __forceinline__ uint64_t dtw_group_base(const uint64_t *__restrict__ off, int i) {
  return (off[i] >> 6) + (uint64_t)i;
}

// One level = three launches.
//   k_dtw_values  ONE workgroup, four wavefronts: the serial recurrence and nothing else.  Rows in lanes, strips of 64
//                 rows, lane l works on column j - l; per step two DPP shifts, three additions, two minima.  The values
//                 D[i][j] go to a band with the distances' layout (one 8-cell store per lane and 8 steps).
//   k_dtw_codes   one workgroup per strip, all at once: the predecessor of every cell from D and the distances (the
//                 first of the candidates (i-1,j), (i,j-1), (i-1,j-1) whose sum equals D[i][j]: fastdtw's order), and
//                 for the cells of the strip's last row the column at which the best path entered the strip.
//   k_dtw_trace   ONE workgroup: one thread hops from strip to strip over the entry columns, every strip is then
//                 walked by its own lane, the segments are copied to their places; its tail computes the next
//                 (finer) level's windows.
// Until round 3 the recurrence kernel also produced the codes and carried the entry columns along: ~32 instructions
// per step on the wavefront that owns a strip, 170-200 cycles per step.  Neither is on the dependence chain.
struct dtw_level_args {
  int len_x, len_y;
  const int32_t *lo, *hi;       // (lo_w, hi_w, off_w alias them: the trace's tail rewrites the tables)
  const uint64_t *off;          // rows' offsets (cells of the windows): where a row's predecessor planes are
  const uint64_t *soff;         // strips' offsets in the skewed bands
  uint64_t cap_rows, cap_skew;
  double *dist;                 // skewed band of distances; the strips' last rows get their cells' entry columns
  double *dval;                 // skewed band of D values
  uint64_t *predm;
  double *bnd_global;           // DTW_WAVES x (len_y + 2), or null: boundary rows in LDS
  int32_t *path, *rev, *sinfo;  // sinfo: 3 x strips
  int64_t *path_len;
  double *out_dist;
  int *status;                  // [0] band overflow (k_dtw_dist), [1] this level has no entry columns
  long long *dbg;
  int lds_bytes;
  int next_len_x, next_len_y, radius;
  int32_t *lo_w, *hi_w;
  uint64_t *off_w, *soff_w;
  const double *x, *y;          // this level's series, or null (k_dtw_small: distances computed by the kernel itself)
  int dim;
};

// ---- batches of pairs ------------------------------------------------------------------------------------
// Every kernel of the level recursion takes a batch of up to KWY_BATCH_MAX pairs (descriptors by value in the kernel
// arguments, as for the analysis kernels) and the level to work on; workgroup (.., pair) builds the pair's
// dtw_level_args itself.  All pairs share one scratch layout, sized by the longest series of the batch: the pieces of
// pair p live at scratch + p * stride + off_*.  A pair has its own number of levels (lengths halve until one is
// shorter than radius + 2); the launches go from the batch's coarsest level down to 0 and a pair that has no such
// level yet sits the launch out.  A single call is a batch of one: the same kernels everywhere.
#define DTW_OFFLV 32      // (lengths below 2^30: fewer levels than this)
struct dtw_pair {
  const double *x, *y;
  int Tx, Ty;
  double *dist;
  int32_t *path;
  int64_t *path_len;
};
struct dtw_batch {
  int n, dim, radius;
  int lds_bytes;
  char *scratch;
  uint64_t stride;
  uint64_t cap_rows, cap_skew;
  uint64_t off_x[DTW_OFFLV], off_y[DTW_OFFLV];                // coarsened series of level l >= 1
  uint64_t off_dist, off_dval, off_soff, off_predm, off_lo, off_hi, off_off, off_bnd, off_pathA, off_pathB, off_rev,
      off_sinfo, off_lenA, off_lenB, off_status;
  int bnd_lds;                  // boundary rows of the strips in LDS (else off_bnd)
  long long *dbg;
  dtw_pair p[KWY_BATCH_MAX];
};
__host__ __device__ static inline int dtw_num_levels(int Tx, int Ty, int radius) {
  int n = 1;
  while (!(Tx < radius + 2 || Ty < radius + 2)) { Tx /= 2; Ty /= 2; ++n; }
  return n;
}
// the arguments of pair `pi` at level `lev` (0 = the finest); false: the pair has no such level
__device__ __forceinline__ bool dtw_make_args(const dtw_batch &b, int pi, int lev, bool with_series, dtw_level_args &a) {
  const dtw_pair &q = b.p[pi];
  const int nlev = dtw_num_levels(q.Tx, q.Ty, b.radius);
  if (lev >= nlev) return false;
  char *base = b.scratch + (uint64_t)pi * b.stride;
  a.len_x = q.Tx >> lev; a.len_y = q.Ty >> lev;
  a.lo = (int32_t *)(base + b.off_lo); a.hi = (int32_t *)(base + b.off_hi);
  a.off = (uint64_t *)(base + b.off_off); a.soff = (uint64_t *)(base + b.off_soff);
  a.cap_rows = b.cap_rows; a.cap_skew = b.cap_skew;
  a.dist = (double *)(base + b.off_dist); a.dval = (double *)(base + b.off_dval);
  a.predm = (uint64_t *)(base + b.off_predm);
  a.bnd_global = b.bnd_lds ? nullptr : (double *)(base + b.off_bnd);
  const bool top = lev == 0;
  a.path = top ? q.path : (int32_t *)(base + ((lev & 1) ? b.off_pathA : b.off_pathB));
  a.path_len = top ? q.path_len : (int64_t *)(base + ((lev & 1) ? b.off_lenA : b.off_lenB));
  a.rev = (int32_t *)(base + b.off_rev); a.sinfo = (int32_t *)(base + b.off_sinfo);
  a.out_dist = q.dist;
  a.status = (int *)(base + b.off_status);
  a.dbg = b.dbg;
  a.lds_bytes = b.lds_bytes;
  a.next_len_x = top ? 0 : q.Tx >> (lev - 1);
  a.next_len_y = top ? 0 : q.Ty >> (lev - 1);
  a.radius = b.radius;
  a.lo_w = (int32_t *)(base + b.off_lo); a.hi_w = (int32_t *)(base + b.off_hi);
  a.off_w = (uint64_t *)(base + b.off_off); a.soff_w = (uint64_t *)(base + b.off_soff);
  a.x = with_series ? (lev == 0 ? q.x : (const double *)(base + b.off_x[lev])) : nullptr;
  a.y = with_series ? (lev == 0 ? q.y : (const double *)(base + b.off_y[lev])) : nullptr;
  a.dim = b.dim;
  return true;
}

#define DTW_WAVES 4
// Steps between two looks at the previous strip's progress.  Strip k+1 trails strip k by the 63
// steps of the row skew plus one chunk.
#define DTW_CHUNK 16
#ifndef DTW_RING
#define DTW_RING 5           // chunks of distances held in registers (all but one of them in flight)
#endif
#ifndef DTW_BLK
#define DTW_BLK 16           // steps between two rounds of bookkeeping (stores of D, boundary row, progress word)
#endif
typedef double dtw_d2 __attribute__((ext_vector_type(2), aligned(16)));

template <bool BND_LDS>
__device__ __forceinline__ void dtw_values_body(const dtw_level_args &a, unsigned char *bt /* boundary rows (BND_LDS) */) {
  const long long t_start = a.dbg ? clock64() : 0;
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
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform: a scalar)
  if (a.status[0] != 0) return;
  const int len_x = a.len_x, len_y = a.len_y;
  const int32_t *lo = a.lo, *hi = a.hi;
  const uint64_t *off = a.off;
  long long *dbg = a.dbg;
  const double INF = INFINITY;
  const int rowlen = len_y + 2;  // boundary rows are indexed by j + 1 (entry 0 is column -1)
  double *const lds_rows = (double *)bt;
  double *const bnd_global = a.bnd_global;
#define BROW(buf, ix) (BND_LDS ? lds_rows[(buf) * rowlen + (ix)] : bnd_global[(size_t)(buf) * rowlen + (ix)])
  if (threadIdx.x < DTW_WAVES) DTW_PROG_STORE(threadIdx.x, -1);
  if (dbg && lane == 0) dbg[16 + wv] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID: where the wave runs
  __syncthreads();
  const int nstrips = (len_x + 63) / 64;
  for (int k = wv; k < nstrips; k += DTW_WAVES) {
    const int i0 = k * 64;
    const int i = i0 + lane;
    const bool valid = i < len_x;
    const int rl = valid ? lo[i] : 0, rh = valid ? hi[i] : -1;
    const int rw = rh - rl;  // last valid index of the row
    const int ilast = min(i0 + 63, len_x - 1);
    const int L = ilast - i0;  // lane of the strip's last row
    // (wave-uniform values through readfirstlane: the tables are not restrict-qualified -- the trace's tail rewrites
    // them -- and a uniform value left in a vector register turns every test on it into vector compares)
    const int jmin = __builtin_amdgcn_readfirstlane(lo[i0]), jmax = __builtin_amdgcn_readfirstlane(hi[ilast]);
    // the previous strip's last row: where its boundary values are valid
    const int plo = k > 0 ? __builtin_amdgcn_readfirstlane(lo[i0 - 1]) : 0;
    const int phi = k > 0 ? __builtin_amdgcn_readfirstlane(hi[i0 - 1]) : -1;
    const int pbuf = (k + DTW_WAVES - 1) % DTW_WAVES, nbuf = k % DTW_WAVES;
    // both bands: the strip's base plus the lane; two steps per 16-byte element, 64 elements per pair of steps
    const uint64_t sbv = a.soff[k];
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sbv >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbv);
    const dtw_d2 *dbase = (const dtw_d2 *)(a.dist + sb) + lane;
    dtw_d2 *vbase = (dtw_d2 *)(a.dval + sb) + lane;
    double v1 = INF;      // this lane's value at the previous step
    double up_prev = INF; // the `up` input of the previous step = this step's diagonal input
    const int nsteps = (jmax - jmin + 1) + 63, nsteps16 = (nsteps + 15) & ~15;
    if (dbg && lane == 0 && k < 48) { dbg[64 + 4 * k] = clock64() - t_start; dbg[64 + 4 * k + 2] = nsteps; dbg[64 + 4 * k + 3] = jmin; }
    const int shift = lane + rl - jmin;  // this lane's row index at step s is s - shift
    // the last row's range of steps and last column, as scalars: its boundary writes are guarded on the scalar unit
    const int shL = __builtin_amdgcn_readlane(shift, L), rwL = __builtin_amdgcn_readlane(rw, L);
    const int rhL = __builtin_amdgcn_readlane(rh, L);
    // blocks of DTW_BLK steps (s0 a multiple of it) that lie wholly inside the last row: s0 in [sIn0, sIn0 + inSpan]
    int sIn0 = (shL + DTW_BLK - 1) & ~(DTW_BLK - 1);
    const int sIn1 = (shL + rwL - (DTW_BLK - 1)) & ~(DTW_BLK - 1);
    if (sIn1 < sIn0) sIn0 = 0x40000000;
    const unsigned inSpan = sIn1 >= sIn0 ? (unsigned)(sIn1 - sIn0) : 0u;
    // lane 0's diagonal input at the first step: D[i0-1][jmin-1]
    if (k == 0 && lane == 0 && jmin == 0) up_prev = 0.0;  // D[-1][-1] = 0: the origin of the recurrence
    // Distances of this lane's row: a ring of DTW_RING chunks of 16 steps in registers, each fetched
    // DTW_RING - 1 chunks before it is used -- the band was written by another kernel on
    // other XCDs and comes from memory (about 3500 cycles on an otherwise idle chip), and a step is
    // only ~50 cycles.  (A fetch beyond the strip's last chunk repeats that chunk: nobody uses it.)
    double ring[DTW_RING][DTW_CHUNK];
    auto fetch = [&](double (&b)[DTW_CHUNK], int cstart) {
      const dtw_d2 *p = dbase + (size_t)(min(cstart, nsteps16 - DTW_CHUNK) >> 1) * 64;
#pragma unroll
      for (int q = 0; q < DTW_CHUNK / 2; ++q) {
        const dtw_d2 v = p[q * 64];
        b[2 * q] = v.x;
        b[2 * q + 1] = v.y;
      }
    };
#pragma unroll
    for (int r = 0; r < DTW_RING - 1; ++r) fetch(ring[r], DTW_CHUNK * r);
    bool first_chunk = true;
    // the boundary values lane 0 needs in the DTW_CHUNK steps from c0 on: lane t holds the one of step c0 + t (the
    // vector is rotated by one lane per step, so that lane 0 always holds the current one)
    auto load_brot = [&](int c0) {
      const int jb = jmin + c0 + lane;
      double b = INF;
      if (jb >= plo && jb <= phi) b = BROW(pbuf, jb + 1);
      return b;
    };
    auto covered = [&](long long pv, int need) {     // (every lane holds the same word: tests on the scalar unit)
      const int ps = __builtin_amdgcn_readfirstlane((int)(pv >> 32));
      const int pc = __builtin_amdgcn_readfirstlane((int)(pv & 0xffffffffll)) - 1;
      return ps > k - 1 || (ps == k - 1 && pc >= need);
    };
    // The progress word and the boundary values of the NEXT chunk are read before the steps of the current one
    // (an LDS round trip is ~130 cycles, two of them one after the other a fifth of a chunk): if the word read
    // then already covered the columns, the values read behind it are the final ones.
    long long pv_ahead = k > 0 ? DTW_PROG_LOAD(pbuf) : 0;
    double brot_ahead = k > 0 ? load_brot(0) : INF;
    // one chunk of DTW_CHUNK steps on `cur`; `fill` (the buffer used one chunk ago) is refilled meanwhile
    auto chunk = [&](int c0, const double (&cur)[DTW_CHUNK], double (&fill)[DTW_CHUNK]) {
      fetch(fill, c0 + DTW_CHUNK * (DTW_RING - 1));
      double brot = brot_ahead;
      if (k > 0) {
        // wait until the previous strip's last row has produced the columns this chunk reads
        const int need = min(jmin + c0 + DTW_CHUNK - 1, phi);  // last column we may read (valid ones only)
        if (!covered(pv_ahead, need)) {
          while (!covered(DTW_PROG_LOAD(pbuf), need)) __builtin_amdgcn_s_sleep(1);
          DTW_ACQUIRE();
          brot = load_brot(c0);
        }
        if (first_chunk) {
          const int jd = jmin - 1;
          const double dv = (jd >= plo && jd <= phi) ? BROW(pbuf, jd + 1) : INF;
          if (lane == 0) up_prev = dv;
          first_chunk = false;
        }
        pv_ahead = DTW_PROG_LOAD(pbuf);
        DTW_ACQUIRE();
        brot_ahead = load_brot(c0 + DTW_CHUNK);
      }
#pragma unroll
      for (int blk = 0; blk < DTW_CHUNK / DTW_BLK; ++blk) {
        const int s0 = c0 + DTW_BLK * blk;
        double hist[DTW_BLK];
#pragma unroll
        for (int u = 0; u < DTW_BLK; ++u) {
          // up = D[i-1][j]: the neighbouring lane's value of the previous step (lane 0: the boundary row).
          // The diagonal D[i-1][j-1] is what `up` was one step ago -- no second shift.
          const double nbrot = dtw_wave_rol1(brot);
          const double up = dtw_wave_shr1(v1, brot);   // brot dies here: the shift lands in its register
          brot = nbrot;
          const double dg = up_prev;
          up_prev = up;
          // D = min(up + d, left + d, diagonal + d) = min(up, left, diagonal) + d, bit for bit: rounding is monotone
          // (x <= y => fl(x + d) <= fl(y + d)), so the smallest sum is the sum of the smallest.  One addition instead
          // of three; which candidate attains the minimum, in fastdtw's order and on the three sums, is k_dtw_codes'
          // business.  A lane outside its row adds +inf.
          const double best = dtw_min_raw(up, dtw_min_raw(v1, dg)) + cur[DTW_BLK * blk + u];
          hist[u] = best;
          v1 = best;
        }
        // the values go to the band of D values, same places as the distances (a lane outside its row: +inf)
        {
          dtw_d2 *p = vbase + (size_t)(s0 >> 1) * 64;
#pragma unroll
          for (int q = 0; q < DTW_BLK / 2; ++q) p[q * 64] = dtw_d2{hist[2 * q], hist[2 * q + 1]};
        }
        // The strip's last row goes to the boundary buffer (its lane only, the steps inside its row only), and how
        // far it has got is published after the writes.  A lone wavefront on its SIMD issues one instruction of ANY
        // kind per ~5 cycles: the scalar bookkeeping here counts like the arithmetic, so the common case (the whole
        // block inside the row) is one wave-uniform test and one masked region.
        if ((unsigned)(s0 - sIn0) <= inSpan) {            // wave-uniform: all steps inside the last row
          DTW_RELEASE();
          if (lane == L) {
#pragma unroll
            for (int u = 0; u < DTW_BLK; ++u) BROW(nbuf, jmin + s0 - L + 1 + u) = hist[u];
            DTW_PROG_STORE(nbuf, ((long long)k << 32) | (long long)(unsigned int)(jmin + s0 + DTW_BLK - L));
          }
        } else if ((unsigned)(s0 - (shL - (DTW_BLK - 1))) <= (unsigned)(rwL + (DTW_BLK - 1))) {   // straddles an end
          DTW_RELEASE();
          if (lane == L) {
#pragma unroll
            for (int u = 0; u < DTW_BLK; ++u) {
              const int s = s0 + u;
              if (s >= shL && s <= shL + rwL) BROW(nbuf, jmin + s - L + 1) = hist[u];
            }
            const int jdone = min(jmin + (s0 + DTW_BLK - 1) - L, rhL);
            DTW_PROG_STORE(nbuf, ((long long)k << 32) | (long long)(unsigned int)(jdone + 1 > 0 ? jdone + 1 : 0));
          }
        }
      }
    };
    // whole chunks (the steps behind nsteps see +inf only), the ring's phases unrolled
    for (int c0 = 0; c0 < nsteps; c0 += DTW_RING * DTW_CHUNK) {
#pragma unroll
      for (int ph = 0; ph < DTW_RING; ++ph) {
        if (ph > 0 && c0 + ph * DTW_CHUNK >= nsteps) break;
        chunk(c0 + ph * DTW_CHUNK, ring[ph], ring[(ph + DTW_RING - 1) % DTW_RING]);
      }
    }
    DTW_RELEASE();
    if (lane == L) DTW_PROG_STORE(nbuf, ((long long)k << 32) | 0x7fffffffll);
    if (dbg && lane == 0 && k < 48) dbg[64 + 4 * k + 1] = clock64() - t_start;
  }
#undef BROW
  if (dbg) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long t = clock64() - t_start;
      atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)t);
      dbg[2] = t;
    }
  }
}

// Codes and entry columns of one strip.
// Phase 1 works in the recurrence's own coordinates (lane = row, step by step: every load is coalesced): the
// predecessor of a cell is the first of (i-1, j), (i, j-1), (i-1, j-1) whose value plus the cell's distance equals
// D[i][j] (fastdtw's min() keeps the first of equal candidates; the sums are the ones the recurrence formed).  A lane
// collects the bits of its row over 64 steps -- 64 consecutive columns -- and ORs them into the row's planes.
// Phase 2, a wavefront per row: for every cell the place in the row above where the best path into the cell comes
// from (the nearest cell at or left of it whose predecessor is not (i, j-1), and that cell's own predecessor).
// Phase 3, a thread per cell of the strip's last row: follow those places up through the strip's rows, all of them in
// LDS.  A strip whose planes and places do not fit there builds the planes in memory and flags the level: the trace
// then walks the path in one piece.
template <int DTW_CODES_NT>
__device__ __forceinline__ void dtw_codes_body(const dtw_level_args &a, int k, unsigned char *bt) {
  __shared__ int s_lo[64], s_w[64], s_po[65], s_gb[65];
  if (a.status[0] != 0) return;
  const int len_x = a.len_x;
  const int32_t *lo = a.lo, *hi = a.hi;
  const uint64_t *off = a.off;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i0 = 64 * k, ilast = min(i0 + 63, len_x - 1), nrows = ilast - i0 + 1;
  const double INF = INFINITY;
  const uint64_t gb0 = dtw_group_base(off, i0);
  __syncthreads();          // (the tables below may still be in use by the previous strip of a fused launch)
  if (tid < 64) {
    const int r = min(i0 + tid, ilast);
    s_lo[tid] = lo[r];
    s_w[tid] = tid < nrows ? hi[r] - lo[r] + 1 : 0;
    s_gb[tid] = (int)(dtw_group_base(off, r) - gb0);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int r = 0; r < 64; ++r) { s_po[r] = acc; acc += s_w[r]; }
    s_po[64] = acc;
    s_gb[64] = s_gb[nrows - 1] + ((s_w[nrows - 1] - 1) >> 6) + 1;     // groups of the strip
  }
  __syncthreads();
  const int gtot = s_gb[64];
  const size_t m_bytes = (size_t)gtot * 16;
  const bool fits = m_bytes + (size_t)s_po[64] * sizeof(int32_t) <= (size_t)a.lds_bytes;
  uint64_t *M = (uint64_t *)bt;                    // planes of the strip: 2 words per group
  int32_t *P = (int32_t *)(bt + m_bytes);
  uint64_t *Mg = a.predm + 2 * gb0;
  if (fits) { for (int w = tid; w < 2 * gtot; w += DTW_CODES_NT) M[w] = 0ull; }
  else      { for (int w = tid; w < 2 * gtot; w += DTW_CODES_NT) Mg[w] = 0ull; }
  __syncthreads();
  // ---- phase 1
  const int jmin = s_lo[0];
  const int nsteps16 = dtw_strip_steps16(lo, hi, len_x, k);
  const uint64_t sb = a.soff[k];
  const bool valid = lane < nrows;
  const int l = s_lo[lane], wdt = s_w[lane];       // this lane's row
  const int groups = valid ? ((wdt - 1) >> 6) + 1 : 0;
  // lane 0's (i-1, j): the previous strip's last row, lane 63 of that strip's band
  const int plo = k > 0 ? lo[i0 - 1] : 0, phi = k > 0 ? hi[i0 - 1] : -1;
  const int pjmin = k > 0 ? lo[i0 - 64] : 0;
  const uint64_t psb = k > 0 ? a.soff[k - 1] : 0;
  auto phase1 = [&](uint64_t *planes) {
    // blocks of 16 steps: enough of them for all the wavefronts, one batch of loads each
    for (int b = wv; 16 * b < nsteps16; b += DTW_CODES_NT / 64) {
      const int sb0 = 16 * b;
      // the boundary values of the block's steps: lane t < 16 holds D[i0-1][jmin + sb0 + t]
      double bvec = INF;
      {
        const int jb = jmin + sb0 + lane;
        if (lane < 16 && jb >= plo && jb <= phi) bvec = a.dval[dtw_skew_index(psb, jb - pjmin + 63, 63)];
      }
      double prevD = sb0 > 0 ? a.dval[dtw_skew_index(sb, sb0 - 1, lane)] : INF;
      double Dc[16], dt[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        Dc[u] = a.dval[dtw_skew_index(sb, sb0 + u, lane)];
        dt[u] = a.dist[dtw_skew_index(sb, sb0 + u, lane)];
      }
      uint32_t w0 = 0, w1 = 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double bfirst = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bvec), u),
                                               __builtin_amdgcn_readlane(__double2loint(bvec), u));
        const double up = dtw_wave_shr1(prevD, bfirst);
        const double left = prevD;
        const int c = (jmin + sb0 + u - lane) - l;     // the cell's place in its row's window
        const bool in = valid && c >= 0 && c < wdt;
        const bool e0 = in && (up + dt[u] == Dc[u]), e1 = in && (left + dt[u] == Dc[u]);
        w0 |= (uint32_t)e0 << u;
        w1 |= (uint32_t)e1 << u;
        prevD = Dc[u];
      }
      // bit u of the words = column c0 + u of the row's window
      const int c0 = (jmin + sb0 - lane) - l;
      const int g = c0 >> 6, sh = c0 & 63;
      const uint64_t lo0 = (uint64_t)w0 << sh, lo1 = (uint64_t)w1 << sh;
      const uint64_t hi0 = sh > 48 ? (uint64_t)w0 >> (64 - sh) : 0ull, hi1 = sh > 48 ? (uint64_t)w1 >> (64 - sh) : 0ull;
      uint64_t *row = planes + 2 * (size_t)s_gb[lane];
      if (g >= 0 && g < groups) {
        if (lo0) atomicOr((unsigned long long *)&row[2 * g], (unsigned long long)lo0);
        if (lo1) atomicOr((unsigned long long *)&row[2 * g + 1], (unsigned long long)lo1);
      }
      if (g + 1 >= 0 && g + 1 < groups) {
        if (hi0) atomicOr((unsigned long long *)&row[2 * (g + 1)], (unsigned long long)hi0);
        if (hi1) atomicOr((unsigned long long *)&row[2 * (g + 1) + 1], (unsigned long long)hi1);
      }
    }
  };
  if (!fits) {
    phase1(Mg);
    if (tid == 0) atomicExch(&a.status[1], 1);
    return;
  }
  phase1(M);
  __syncthreads();
  // ---- phase 2: the planes go to memory, the places to LDS
  for (int w = tid; w < 2 * gtot; w += DTW_CODES_NT) Mg[w] = M[w];
  for (int rr = wv; rr < nrows; rr += DTW_CODES_NT / 64) {
    const int rl = s_lo[rr], w = s_w[rr];
    const uint64_t *row = M + 2 * (size_t)s_gb[rr];
    int32_t *Pr = P + s_po[rr];
    int carry = rl - 1;                      // the place of the last cell seen whose predecessor is not (i, j-1)
    for (int c0 = 0, g = 0; c0 < w; c0 += 64, ++g) {
      const uint64_t m0 = row[2 * g], m1 = row[2 * g + 1];
      const uint64_t mi = w - c0 >= 64 ? ~0ull : (~0ull >> (64 - (w - c0)));
      const uint64_t vm = (m0 | ~m1) & mi;              // cells that do not copy their left neighbour
      const uint64_t below = vm & (~0ull >> (63 - lane));
      int place;                                        // a column of the row above (or of the boundary row)
      if (below) {
        const int pos = 63 - __clzll((long long)below);
        place = (rl + c0 + pos) - (int)(((m0 >> pos) & 1ull) ? 0 : 1);
      } else {
        place = carry;
      }
      if (c0 + lane < w) Pr[c0 + lane] = place;
      if (vm) {
        const int pos = 63 - __clzll((long long)vm);
        carry = (rl + c0 + pos) - (int)(((m0 >> pos) & 1ull) ? 0 : 1);
      }
    }
  }
  __syncthreads();
  // ---- phase 3: the last row's cells, up through the rows; a place outside a row's window belongs to an
  //      unreachable cell.  The entry column of cell (last row, j) is kept in the cell's own (dead) distance.
  const int wl = s_w[nrows - 1], ll = s_lo[nrows - 1], L = nrows - 1;
  for (int c = tid; c < wl; c += DTW_CODES_NT) {
    int col = P[s_po[L] + c];
    for (int rr = L - 1; rr >= 0; --rr) {
      const int cc = min(max(col - s_lo[rr], 0), s_w[rr] - 1);
      col = P[s_po[rr] + cc];
    }
    *(int32_t *)&a.dist[dtw_skew_index(sb, (ll + c) - jmin + L, L)] = col;
  }
}
#define DTW_CODES_THREADS 1024

// The back-trace of one level and the next level's windows: ONE workgroup.
#define DTW_TRACE_NT 256
__device__ __forceinline__ void dtw_trace_body(const dtw_level_args &a, unsigned char *bt) {
  __shared__ int s_n;
  const int len_x = a.len_x, len_y = a.len_y;
  const int32_t *lo = a.lo, *hi = a.hi;
  const uint64_t *off = a.off;
  long long *dbg = a.dbg;
  if (a.status[0] != 0) {
    if (threadIdx.x == 0) { *a.path_len = 0; *a.out_dist = NAN; }
    // keep the next level's tables defined (full windows: its distance kernel sees the overflow and leaves the status)
    if (a.next_len_x > 0)
      dtw_window_scan_body<DTW_TRACE_NT>((const int32_t *)nullptr, 0, a.radius, a.next_len_x, a.next_len_y, a.lo_w,
                                         a.hi_w, a.off_w, a.soff_w, a.cap_rows, a.cap_skew, a.status, bt,
                                         (size_t)a.lds_bytes);
    return;
  }
  const long long t_dp = dbg ? clock64() : 0;
  const bool serial = a.status[1] != 0;       // no entry columns: the whole path is one segment
  const int nstrips = serial ? 1 : (len_x + 63) / 64;
  if (threadIdx.x == 0) {
    const int ks = (len_x - 1) / 64, Ls = (len_x - 1) - 64 * ks;
    *a.out_dist = a.dval[dtw_skew_index(a.soff[ks], (len_y - 1) - lo[64 * ks] + Ls, Ls)];
  }
  // ---- (1) one thread hops from strip to strip: the path leaves strip k through
  //      (last row, exitc[k]) and the entry column stored there is where it leaves strip k-1.
  //      (2) every strip is walked by its own lane, all at once, over the predecessor planes.
  //      (3) the strips' cell counts are summed, (4) the segments are copied to their places.
  // A walk is a chain of dependent loads (row window -> group address -> planes), ~1300 cycles per cell
  // from global memory: the rows' windows are staged in LDS first, and after the hop -- which says between which
  // columns each strip's segment runs -- the groups of the planes that cover those columns (two or three per row
  // instead of the whole window), when they fit.
  int32_t *exitc = a.sinfo, *sbase = a.sinfo + nstrips, *cnt = a.sinfo + 2 * nstrips;
  const size_t lds_bytes = (size_t)a.lds_bytes;
  const size_t tbl_bytes = ((8ull * (size_t)len_x + 15) & ~15ull) + 8ull * DTW_TRACE_NT + 8ull * ((size_t)len_x + 2);
  const bool tbl_ok = lds_bytes >= tbl_bytes + 64;
  int32_t *t_lo = (int32_t *)bt, *t_hi = t_lo + len_x;
  uint64_t *t_tot = (uint64_t *)(bt + ((8ull * (size_t)len_x + 15) & ~15ull));
  uint64_t *t_so = t_tot + DTW_TRACE_NT;              // first staged group of every row
  uint64_t *t_pm = (uint64_t *)(bt + ((tbl_bytes + 15) & ~15ull));
  if (tbl_ok) {
    for (int i = threadIdx.x; i < len_x; i += DTW_TRACE_NT) { t_lo[i] = lo[i]; t_hi[i] = hi[i]; }
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
        e = *(const int32_t *)&a.dist[dtw_skew_index(a.soff[k], c - LO[i0] + (il - i0), il - i0)];
        e = min(max(e, 0), cj);
      }
      sbase[k] = base;
      base += (il - i0 + 1) + (cj - e) + 1;   // rows + columns: more cells than the strip can hold
      cj = e;
    }
  };
  if (threadIdx.x == 0) {
    if (serial) { exitc[0] = len_y - 1; sbase[0] = 0; }
    else if (tbl_ok) hop(t_lo, t_hi);
    else hop(lo, hi);
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[21] = clock64() - t_dp;
  // the groups row i's part of the segment can touch: columns [entry of its strip, exit of its strip] of its window
  auto first_group = [&](int i, int l) { const int k = serial ? 0 : i >> 6; const int e = k > 0 ? exitc[k - 1] : 0; return (max(l, e) - l) >> 6; };
  auto last_group = [&](int i, int l, int h) { const int k = serial ? 0 : i >> 6; return (max(min(h, exitc[k]), l) - l) >> 6; };
  bool pm_ok = false;
  if (tbl_ok) {
    kwy_block_count_scan<DTW_TRACE_NT>([&](int64_t i) -> uint64_t {
      const int l = t_lo[i], h = t_hi[i];
      return (uint64_t)(max(last_group((int)i, l, h) - first_group((int)i, l), 0) + 1);
    }, len_x, t_so, t_tot);
    __syncthreads();
    pm_ok = t_so[len_x] * 16 + ((tbl_bytes + 15) & ~15ull) <= lds_bytes;
    if (pm_ok) {
      // four rows per thread at a time, all their loads in flight together (two round trips to memory per batch:
      // the row's offset, then its groups; a row has rarely more than three)
      for (int ib = threadIdx.x; ib < len_x; ib += 4 * DTW_TRACE_NT) {
        const uint64_t *src[4];
        uint64_t *dst[4];
        int n[4];
        uint64_t v[4][3][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ib + r * DTW_TRACE_NT;
          n[r] = 0;
          if (i < len_x) {
            const int l = t_lo[i], h = t_hi[i];
            const int g0 = first_group(i, l), g1 = max(last_group(i, l, h), g0);
            n[r] = g1 - g0 + 1;
            src[r] = a.predm + 2 * (dtw_group_base(off, i) + (uint64_t)g0);
            dst[r] = t_pm + 2 * t_so[i];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int g = 0; g < 3; ++g)
            if (g < n[r]) { v[r][g][0] = src[r][2 * g]; v[r][g][1] = src[r][2 * g + 1]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int g = 0; g < 3; ++g)
            if (g < n[r]) { dst[r][2 * g] = v[r][g][0]; dst[r][2 * g + 1] = v[r][g][1]; }
          for (int g = 3; g < n[r]; ++g) { dst[r][2 * g] = src[r][2 * g]; dst[r][2 * g + 1] = src[r][2 * g + 1]; }
        }
      }
    }
    __syncthreads();
  }
  if (dbg && threadIdx.x == 0) { dbg[24] = clock64() - t_dp; dbg[25] = (pm_ok ? 1 : 0) + (tbl_ok ? 2 : 0); dbg[26] = 0; dbg[27] = tbl_ok ? (long long)t_so[len_x] : -1; }
  auto walk = [&](int k, int i_stop, int i_from, const int32_t *LO, const int32_t *HI) {
    int i = i_from, j = exitc[k], m = 0;
    int32_t *out = a.rev + 2 * (int64_t)sbase[k];
    int crow = -1, l = 0, h = -1, cg = -1;
    uint64_t gb = 0, m0 = 0, m1 = 0;
    while (i >= i_stop) {
      out[2 * m] = i; out[2 * m + 1] = j; ++m;
      if (i == 0 && j == 0) break;
      if (i != crow) { crow = i; l = LO[i]; h = HI[i]; gb = dtw_group_base(off, i); cg = -1; }
      unsigned int pb = 0;
      if (j >= l && j <= h) {
        const int c = j - l, g = c >> 6;
        if (g != cg) { cg = g; m0 = a.predm[2 * (gb + g)]; m1 = a.predm[2 * (gb + g) + 1]; }
        pb = ((m0 >> (c & 63)) & 1ull) ? 0u : (((m1 >> (c & 63)) & 1ull) ? 1u : 2u);
      }
      if (pb == 0) --i; else if (pb == 1) --j; else { --i; --j; }
      if (j < 0) break;
    }
    cnt[k] = m;
  };
  // The same walk on the staged planes, without branches (35 lanes of one wavefront each walk ~130 cells: what
  // costs is the number of instructions per cell and the two dependent LDS reads, row table -> planes).
  auto walk_staged = [&](int k, int i_stop, int i_from) -> bool {
    int i = i_from, j = exitc[k], m = 0;
    const int ek = k > 0 ? exitc[k - 1] : 0;
    int2 *out = (int2 *)a.rev + (int64_t)sbase[k];
    bool ok = true;
    // the row's table entry {lo, hi, first staged group, staged groups}; the row above is fetched one cell ahead
    struct row_t { int l, h, so, ng; };
    auto load_row = [&](int r) { r = max(r, 0); row_t x; x.l = t_lo[r]; x.h = t_hi[r]; x.so = (int)t_so[r]; x.ng = (int)t_so[r + 1] - x.so; return x; };
    row_t cur = load_row(i), nxt = load_row(i - 1);
    while (true) {
      out[m] = make_int2(i, j); ++m;
      const int c = j - cur.l, g = (c >> 6) - ((max(cur.l, ek) - cur.l) >> 6);
      const bool inwin = (unsigned)c <= (unsigned)(cur.h - cur.l);
      ok = ok && (!inwin || (unsigned)g < (unsigned)cur.ng);
      const uint64_t *pm = t_pm + 2 * (cur.so + min(max(g, 0), cur.ng - 1));
      const uint64_t m0 = pm[0], m1 = pm[1];
      const unsigned pb = !inwin || ((m0 >> (c & 63)) & 1ull) ? 0u : (((m1 >> (c & 63)) & 1ull) ? 1u : 2u);
      const bool origin = i == 0 && j == 0;
      const bool up = pb != 1u;
      i -= up;
      j -= pb != 0u;
      cur.l = up ? nxt.l : cur.l; cur.h = up ? nxt.h : cur.h; cur.so = up ? nxt.so : cur.so; cur.ng = up ? nxt.ng : cur.ng;
      nxt = load_row(i - 1);
      if (origin || i < i_stop || j < 0) break;
    }
    cnt[k] = m;
    return ok;
  };
  for (int k = threadIdx.x; k < nstrips; k += DTW_TRACE_NT) {
    const int i_stop = serial ? 0 : 64 * k, i_from = serial ? len_x - 1 : min(64 * k + 63, len_x - 1);
    bool done = false;
    if (pm_ok) done = walk_staged(k, i_stop, i_from);
    if (!done) {          // the planes in memory (not staged, or -- not expected -- a segment left its columns)
      if (dbg) atomicAdd((unsigned long long *)&dbg[26], 1ull);
      if (tbl_ok) walk(k, i_stop, i_from, t_lo, t_hi);
      else walk(k, i_stop, i_from, lo, hi);
    }
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[22] = clock64() - t_dp;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int k = 0; k < nstrips; ++k) { const int c = cnt[k]; exitc[k] = n; n += c; }   // exitc: now the output offset
    s_n = n;
    *a.path_len = n;
    a.status[1] = 0;                            // the flag belongs to the level
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
      for (int k = threadIdx.x; k < nstrips; k += DTW_TRACE_NT) {
        s_meta[0][k] = cnt[k]; s_meta[1][k] = sbase[k]; s_meta[2][k] = exitc[k];
      }
      __syncthreads();
      const int n = s_n;
      for (int pos = threadIdx.x; pos < n; pos += DTW_TRACE_NT) {
        int lo_s = 0, hi_s = nstrips - 1;            // last strip whose offset is <= pos
        while (lo_s < hi_s) { const int mid = (lo_s + hi_s + 1) >> 1; if (s_meta[2][mid] <= pos) lo_s = mid; else hi_s = mid - 1; }
        const int c = s_meta[0][lo_s], q = pos - s_meta[2][lo_s];
        ((int2 *)a.path)[pos] = ((const int2 *)a.rev)[(int64_t)s_meta[1][lo_s] + (c - 1 - q)];
      }
    } else {
      for (int k = threadIdx.x; k < nstrips; k += DTW_TRACE_NT) {
        const int c = cnt[k];
        const int2 *in = (const int2 *)a.rev + (int64_t)sbase[k];
        int2 *out = (int2 *)a.path + (int64_t)exitc[k];
        for (int m = 0; m < c; ++m) out[c - 1 - m] = in[m];
      }
    }
  }
  if (dbg && threadIdx.x == 0) {
    const long long t_end = clock64();
    atomicAdd((unsigned long long *)&dbg[1], (unsigned long long)(t_end - t_dp));
    dbg[3] = t_end - t_dp; dbg[4] = s_n;
  }
  // ---- the next (finer) level's windows from this path: lo / hi / off of this level are dead now
  if (a.next_len_x > 0) {
    __syncthreads();    // the path is complete (workgroup scope), the LDS tables are free
    dtw_window_scan_body<DTW_TRACE_NT>(a.path, s_n, a.radius, a.next_len_x, a.next_len_y, a.lo_w, a.hi_w, a.off_w,
                                       a.soff_w, a.cap_rows, a.cap_skew, a.status, bt, (size_t)a.lds_bytes);
  }
}

// A level of at most DTW_SMALL_STRIPS strips: recurrence, codes and trace in ONE launch of one workgroup, one after
// the other (the coarse levels: launches of ~12 us each, most of it the first touch of what the launch before wrote
// on another CU, for a few microseconds of work).  Everything a phase writes is read back by the same CU.  The
// distances keep their own launch: one workgroup would take the units one after the other.
#define DTW_SMALL_STRIPS 1
// The distances of a single-strip level from series that fit the LDS whole (a few dozen frames each): one batch of
// loads, then every cell from LDS -- no launch of its own (~10 us plus the gap) for a microsecond of work.
__device__ __forceinline__ void dtw_dist_small(const dtw_level_args &a, unsigned char *lds) {
  if (a.status[0] != 0) return;
  const int len_x = a.len_x, len_y = a.len_y, dim = a.dim, dp = a.dim | 1;     // odd row stride: no bank conflicts
  double *xs = (double *)lds, *ys = xs + (size_t)len_x * dp;
  for (int e = threadIdx.x; e < len_x * dim; e += 64 * DTW_WAVES) { const int r = e / dim, c = e - r * dim; xs[r * dp + c] = a.x[e]; }
  for (int e = threadIdx.x; e < len_y * dim; e += 64 * DTW_WAVES) { const int r = e / dim, c = e - r * dim; ys[r * dp + c] = a.y[e]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const bool valid = lane < len_x;
  const int jmin = a.lo[0], l = valid ? a.lo[lane] : 0, h = valid ? a.hi[lane] : -1;
  const int steps = (int)(a.soff[1] >> 6);            // one strip: its rectangle starts at 0
  const double *xr = xs + (valid ? lane : 0) * dp;
  for (int s = sub; s < steps; s += DTW_WAVES) {
    const int j = jmin + s - lane;
    double d = INFINITY;
    if (j >= l && j <= h) {
      const double *yr = ys + j * dp;
      double acc = 0.0;
      for (int c = 0; c < dim; ++c) { const double df = xr[c] - yr[c]; acc += df * df; }
      d = sqrt(acc);
    }
    a.dist[dtw_skew_index(0, s, lane)] = d;
  }
}
// ---- the kernels: (batch, level); the pair is the last grid dimension in use -------------------------------------
__global__ __launch_bounds__(KWY_THREADS) void k_dtw_halve_all(dtw_batch b) {
  extern __shared__ double hb[];
  const dtw_pair &q = b.p[blockIdx.y];
  const int levels = dtw_num_levels(q.Tx, q.Ty, b.radius) - 1;
  if (levels < 1) return;
  const int C = 1 << levels, dim = b.dim;
  const int chunks0 = (q.Tx + C - 1) / C;
  if ((int)blockIdx.x >= chunks0 + (q.Ty + C - 1) / C) return;
  char *base = b.scratch + (uint64_t)blockIdx.y * b.stride;
  const int s = (int)blockIdx.x < chunks0 ? 0 : 1;
  const int c = s == 0 ? blockIdx.x : blockIdx.x - chunks0;
  const int n = s == 0 ? q.Tx : q.Ty;
  const double *src = s == 0 ? q.x : q.y;
  double *A = hb, *B = hb + (size_t)C * dim;
  const int64_t f0 = (int64_t)c * C;
  const int have = (int)min((int64_t)C, (int64_t)n - f0);
  for (int e = threadIdx.x; e < have * dim; e += KWY_THREADS) A[e] = src[f0 * dim + e];
  __syncthreads();
  for (int l = 1; l <= levels; ++l) {
    const int nl = n >> l, per = C >> l;
    const int i0 = c * per;
    const int cnt = min(per, nl - i0);           // level-l elements of this workgroup (may be <= 0)
    double *out = (double *)(base + (s == 0 ? b.off_x[l] : b.off_y[l]));
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
// the coarsest level of every pair that has `lev` levels below it
__global__ __launch_bounds__(DTW_WS_NT) void k_dtw_window_scan(dtw_batch b, int lev) {
  __shared__ uint64_t tot[DTW_WS_NT];
  dtw_level_args a;
  const dtw_pair &q = b.p[blockIdx.x];
  if (lev != dtw_num_levels(q.Tx, q.Ty, b.radius) - 1 || !dtw_make_args(b, blockIdx.x, lev, false, a)) return;
  dtw_window_scan_first(b.radius, a.len_x, a.len_y, a.lo_w, a.hi_w, a.off_w, a.soff_w, a.cap_rows, a.cap_skew, a.status,
                        tot, sizeof(tot));
}
__global__ __launch_bounds__(KWY_THREADS) void k_dtw_dist(dtw_batch b, int lev) {
  __shared__ double tiles[(64 + 80) * DTW_DTP];
  dtw_level_args a;
  if (!dtw_make_args(b, blockIdx.y, lev, true, a)) return;
  dtw_dist_body(a.x, a.y, a.dim, a.len_x, a.len_y, a.lo, a.hi, a.soff, a.dist, a.status, blockIdx.x, gridDim.x,
                (unsigned char *)tiles);
}
template <bool BND_LDS>
__global__ __launch_bounds__(64 * DTW_WAVES) void k_dtw_values(dtw_batch b, int lev) {
  extern __shared__ unsigned char bt[];
  dtw_level_args a;
  if (!dtw_make_args(b, blockIdx.x, lev, false, a)) return;
  dtw_values_body<BND_LDS>(a, bt);
}
__global__ __launch_bounds__(DTW_CODES_THREADS) void k_dtw_codes(dtw_batch b, int lev) {
  extern __shared__ unsigned char bt[];
  dtw_level_args a;
  if (!dtw_make_args(b, blockIdx.y, lev, false, a)) return;
  if ((int)blockIdx.x >= (a.len_x + 63) / 64) return;
  dtw_codes_body<DTW_CODES_THREADS>(a, blockIdx.x, bt);
}
__global__ __launch_bounds__(DTW_TRACE_NT) void k_dtw_trace(dtw_batch b, int lev) {
  extern __shared__ unsigned char bt[];
  dtw_level_args a;
  if (!dtw_make_args(b, blockIdx.x, lev, false, a)) return;
  dtw_trace_body(a, bt);
}
// (with_dist: the distances from series staged in LDS -- the host has checked that every pair's fit)
__global__ __launch_bounds__(64 * DTW_WAVES) void k_dtw_small(dtw_batch b, int lev, int with_dist) {
  extern __shared__ unsigned char bt[];
  dtw_level_args a;
  if (!dtw_make_args(b, blockIdx.x, lev, with_dist != 0, a)) return;
  if (a.x) { dtw_dist_small(a, bt); __syncthreads(); }
  dtw_values_body<true>(a, bt);
  __syncthreads();
  const int nstrips = (a.len_x + 63) / 64;
  for (int k = 0; k < nstrips; ++k) dtw_codes_body<64 * DTW_WAVES>(a, k, bt);
  __syncthreads();
  dtw_trace_body(a, bt);
}

// ---- host side -----------------------------------------------------------------------------
// Capacities.  Both are BOUNDS, not estimates (until round 4 they were estimates that a pair with a length ratio
// beyond ~4 overflowed: the device entry then returned an empty path).  With (lx, ly) the lengths of a level, r the
// radius and the coarser level's path (monotone, at most one step in either direction per cell):
//   rows' windows   row i takes the columns [2 (jf - r), 2 (jl + r) + 1], jf / jl the path's first / last column in
//                   the coarse rows i/2 - r / i/2 + r: width 2 ext(i/2 - r, i/2 + r) + 4 r + 2, ext = the path's
//                   column extent over those rows.  A coarse row's (or row step's) part of the extent is counted by
//                   at most 2 r + 1 windows, twice each (two fine rows per coarse row):
//                       sum of widths <= 2 (2 r + 1) ly + (lx + 1) (4 r + 2)
//   skewed bands    strip k spans the columns from lo[64 k] to hi[64 k + 63]: 2 ext(32 k - r, 32 k + 31 + r) + 4 r + 2;
//                   a coarse row lies in at most m = (31 + 2 r) / 32 + 1 such ranges:
//                       sum of spans <= m ly + strips (4 r + 2),  plus 63 (skew) + 15 (rounding) steps per strip
// and neither exceeds the full matrix.  (The coarsest level has the full window: lx ly cells with one side shorter
// than r + 2 and the other at most half the finest level's -- below both bounds.)
static uint64_t dtw_cap(int len_x, int len_y, int radius, bool full) {
  const uint64_t worst = (uint64_t)len_x * (uint64_t)len_y;
  if (full) return worst;
  const uint64_t r = (uint64_t)radius;
  const uint64_t bound = 2 * (2 * r + 1) * ((uint64_t)len_y + 2) + ((uint64_t)len_x + 2) * (4 * r + 2);
  return bound < worst ? bound : worst;
}
static uint64_t dtw_cap_skew(int len_x, int len_y, int radius, bool full) {
  const uint64_t strips = ((uint64_t)len_x + 63) / 64;
  const uint64_t worst = 64 * strips * ((uint64_t)len_y + 80);
  if (full) return worst;
  const uint64_t r = (uint64_t)radius, m = (31 + 2 * r) / 32 + 1;
  const uint64_t bound = 64 * (m * ((uint64_t)len_y + 2) + strips * (4 * r + 2 + 63 + 15 + 16));
  return bound < worst ? bound : worst;
}

// the scratch of ONE pair with series of at most (Tx, Ty) frames: the offsets go into `b`, the size is returned
static size_t dtw_layout(int64_t Tx, int64_t Ty, int dim, int radius, bool full, dtw_batch *b) {
  size_t tot = 0;
  auto take = [&](size_t bytes) { const size_t at = tot; tot += kwy_pad(bytes); return (uint64_t)at; };
  dtw_batch tmp;
  if (!b) b = &tmp;
  int lx = (int)Tx, ly = (int)Ty;
  for (int l = 1; l < DTW_OFFLV; ++l) {
    b->off_x[l] = b->off_y[l] = 0;
    if (lx < radius + 2 || ly < radius + 2) continue;
    lx /= 2; ly /= 2;
    b->off_x[l] = take(sizeof(double) * (size_t)lx * dim);
    b->off_y[l] = take(sizeof(double) * (size_t)ly * dim);
  }
  b->off_x[0] = b->off_y[0] = 0;
  b->cap_rows = dtw_cap((int)Tx, (int)Ty, radius, full);
  b->cap_skew = dtw_cap_skew((int)Tx, (int)Ty, radius, full);
  b->off_dist = take(sizeof(double) * b->cap_skew);
  b->off_dval = take(sizeof(double) * b->cap_skew);
  b->off_soff = take(sizeof(uint64_t) * (Tx / 64 + 2));
  b->off_predm = take(16 * (b->cap_rows / 64 + Tx + 8));
  b->off_lo = take(sizeof(int32_t) * Tx);
  b->off_hi = take(sizeof(int32_t) * Tx);
  b->off_off = take(sizeof(uint64_t) * (Tx + 1));
  b->off_bnd = take(sizeof(double) * DTW_WAVES * (Ty + 2));
  b->off_pathA = take(sizeof(int32_t) * 2 * (Tx + Ty + 2));
  b->off_pathB = take(sizeof(int32_t) * 2 * (Tx + Ty + 2));
  b->off_rev = take(sizeof(int32_t) * 2 * (Tx + Ty + Tx / 64 + 8));   // the strips' segments, with slack per strip
  b->off_sinfo = take(sizeof(int32_t) * 3 * (Tx / 64 + 2));
  b->off_lenA = take(64);
  b->off_lenB = take(64);
  b->off_status = take(64);
  return tot;
}

// `pairs`: n <= KWY_BATCH_MAX descriptors with device pointers.  status_out[p]: the pair's status words (device).
static int fastdtw_core(kwy_ctx *ctx, const dtw_pair *pairs, int n, int dim, int radius, bool full, int **status_out) {
  dtw_batch b;
  b.n = n; b.dim = dim; b.radius = radius;
  int64_t Tx = 0, Ty = 0;
  for (int p = 0; p < n; ++p) { b.p[p] = pairs[p]; Tx = std::max<int64_t>(Tx, pairs[p].Tx); Ty = std::max<int64_t>(Ty, pairs[p].Ty); }
  for (int p = n; p < KWY_BATCH_MAX; ++p) b.p[p] = pairs[0];
  b.stride = dtw_layout(Tx, Ty, dim, radius, full, &b);
  b.scratch = (char *)kwy_arena_alloc(ctx, b.stride * (size_t)n);
  if (!b.scratch) { ctx->err = "fastdtw: scratch arena too small"; return KWY_ENOMEM; }
  b.dbg = (long long *)ctx->dbg;
  for (int p = 0; p < n; ++p) status_out[p] = (int *)(b.scratch + (size_t)p * b.stride + b.off_status);
  // levels of the pairs; level l of the batch = level l of every pair that has one
  int nlev[KWY_BATCH_MAX], top = 0;
  for (int p = 0; p < n; ++p) { nlev[p] = dtw_num_levels(pairs[p].Tx, pairs[p].Ty, radius); top = std::max(top, nlev[p]); }
  {
    const int levels = top - 1;
    const size_t halve_lds = levels > 0 ? sizeof(double) * (((size_t)3 << levels) / 2) * dim : 0;
    if (levels > 0 && levels <= DTW_MAXLV && halve_lds <= 64 * 1024) {
      unsigned chunks = 1;
      for (int p = 0; p < n; ++p) {
        const int C = 1 << (nlev[p] - 1);
        chunks = std::max(chunks, (unsigned)((pairs[p].Tx + C - 1) / C + (pairs[p].Ty + C - 1) / C));
      }
      hipLaunchKernelGGL(k_dtw_halve_all, dim3(chunks, n), dim3(KWY_THREADS), halve_lds, ctx->stream, b);
    } else {
      for (int p = 0; p < n; ++p) {              // long series / wide features: level by level
        const double *px = pairs[p].x, *py = pairs[p].y;
        char *base = b.scratch + (size_t)p * b.stride;
        for (int l = 1; l < nlev[p]; ++l) {
          const int lx = pairs[p].Tx >> l, ly = pairs[p].Ty >> l;
          double *cx = (double *)(base + b.off_x[l]), *cy = (double *)(base + b.off_y[l]);
          hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)lx * dim + 255) / 256)), dim3(256), 0, ctx->stream, px,
                             lx, dim, cx);
          hipLaunchKernelGGL(k_dtw_halve, dim3((unsigned)(((size_t)ly * dim + 255) / 256)), dim3(256), 0, ctx->stream, py,
                             ly, dim, cy);
          px = cx; py = cy;
        }
      }
    }
  }
  // the boundary rows of the strips live in LDS when they fit (the other instantiation keeps them in memory)
  const size_t bnd_bytes = sizeof(double) * DTW_WAVES * (Ty + 2);
  b.bnd_lds = bnd_bytes <= 150 * 1024 ? 1 : 0;
  // the codes' places and the back-trace's tables: all a workgroup may have beside the static variables (the kernels
  // take what fits and have a slower way for the rest)
  const size_t big_lds = 150 * 1024;
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_values<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_codes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_trace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_dtw_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));

  for (int l = top - 1; l >= 0; --l) {
    // what the pairs that take part in this level need, at most
    int len_x = 0, len_y = 0;
    bool first = false, small = b.bnd_lds != 0, small_dist = true;
    for (int p = 0; p < n; ++p) {
      if (l >= nlev[p]) continue;
      const int lx = pairs[p].Tx >> l, ly = pairs[p].Ty >> l;
      len_x = std::max(len_x, lx); len_y = std::max(len_y, ly);
      first = first || l == nlev[p] - 1;
      small = small && (lx + 63) / 64 <= DTW_SMALL_STRIPS;
      // (the small kernel takes the distances along when both series fit its LDS)
      small_dist = small_dist && sizeof(double) * (size_t)(lx + ly) * (size_t)(dim | 1) <= 96 * 1024;
    }
    small_dist = small_dist && small;
    if (first)
      hipLaunchKernelGGL(k_dtw_window_scan, dim3(n), dim3(DTW_WS_NT), 0, ctx->stream, b, l);
    const int nstrips = (len_x + 63) / 64;
    // a workgroup per 16 steps of a strip; how many there are is known on the device only: as many workgroups as
    // this level's rectangles can have (the others return at once; more units: the workgroups loop)
    const uint64_t lv_units = std::min(b.cap_skew, dtw_cap_skew(len_x, len_y, radius, full)) / 1024 + 1;
    b.lds_bytes = 0;
    if (!small_dist)
      KWY_PROF(ctx, "k_dtw_dist", hipLaunchKernelGGL(k_dtw_dist, dim3((unsigned)lv_units, n), dim3(KWY_THREADS), 0, ctx->stream, b, l));
    if (small) {
      // A single-strip level asks for the LDS it can use, not for a whole CU's: the series (when it computes the
      // distances itself), the boundary rows, the codes' places and planes of <= 64 rows, the trace's tables -- a few
      // dozen KB for the coarse levels, so that its sixteen workgroups find room beside the resident workgroups of a
      // chip-wide kernel instead of waiting for a CU to drain (0.9 ms per launch inside a step against 0.05 alone).
      const size_t dist_b = small_dist ? sizeof(double) * (size_t)(len_x + len_y) * (size_t)(dim | 1) : 0;
      const size_t bnd_b = sizeof(double) * DTW_WAVES * ((size_t)len_y + 2);
      const size_t groups = 64 * ((size_t)len_y / 64 + 2);                     // 64-column plane groups of 64 rows
      const size_t codes_b = (size_t)64 * 4 * (size_t)std::min<int64_t>(len_y, 8 * (int64_t)radius + 128) + 32 * groups;
      const size_t trace_b = 16 * ((size_t)len_x + 2) + 8 * DTW_TRACE_NT + 16 * groups + 256;
      size_t need = std::max(std::max(dist_b, bnd_b), std::max(codes_b, trace_b)) + 4096;
      need = std::min(big_lds, (need + 1023) & ~(size_t)1023);
      b.lds_bytes = (int)need;
      KWY_PROF(ctx, "k_dtw_small", hipLaunchKernelGGL(k_dtw_small, dim3(n), dim3(64 * DTW_WAVES), need, ctx->stream, b, l,
                                                       small_dist ? 1 : 0));
      continue;
    }
    // the codes kernel asks for the LDS its strips are likely to need (rows x a generous window), not for all of it:
    // several strips then share a CU
    const size_t want = (size_t)64 * 4 * (size_t)std::min<int64_t>(len_y, 8 * (int64_t)radius + 128);
    const size_t codes_lds = std::min(big_lds, std::max<size_t>(want, 16 * 1024));
    if (b.bnd_lds)
      KWY_PROF(ctx, "k_dtw_values", hipLaunchKernelGGL(k_dtw_values<true>, dim3(n), dim3(64 * DTW_WAVES), bnd_bytes, ctx->stream, b, l));
    else
      KWY_PROF(ctx, "k_dtw_values", hipLaunchKernelGGL(k_dtw_values<false>, dim3(n), dim3(64 * DTW_WAVES), 0, ctx->stream, b, l));
    b.lds_bytes = (int)codes_lds;
    KWY_PROF(ctx, "k_dtw_codes", hipLaunchKernelGGL(k_dtw_codes, dim3(nstrips, n), dim3(DTW_CODES_THREADS), codes_lds, ctx->stream, b, l));
    b.lds_bytes = (int)big_lds;
    KWY_PROF(ctx, "k_dtw_trace", hipLaunchKernelGGL(k_dtw_trace, dim3(n), dim3(DTW_TRACE_NT), big_lds, ctx->stream, b, l));
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
  // radius 0: fastdtw 0.3.2 itself fails there (a KeyError: with an odd length the last row gets no window cells)
  if (radius < 1) { ctx->err = "fastdtw: radius must be >= 1"; return KWY_EINVAL; }
  return KWY_OK;
}

extern "C" int kwy_fastdtw_batch_dev(kwy_ctx *ctx, const kwy_dtw_job *jobs, int count, int dim, int radius) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0) { ctx->err = "fastdtw_batch: bad argument"; return KWY_EINVAL; }
  if (count == 0) return KWY_OK;
  int64_t Tx = 0, Ty = 0;
  for (int j = 0; j < count; ++j) {
    KWY_TRY(dtw_check(ctx, jobs[j].x, jobs[j].x_length, jobs[j].y, jobs[j].y_length, dim, radius, jobs[j].dist,
                      jobs[j].path, jobs[j].path_len));
    Tx = std::max(Tx, jobs[j].x_length); Ty = std::max(Ty, jobs[j].y_length);
  }
  KWY_HIP(hipSetDevice(ctx->device));
  // launches of up to KWY_BATCH_MAX pairs, each with its own slice of the arena
  size_t bytes = 0;
  for (int j0 = 0; j0 < count; j0 += KWY_BATCH_MAX) {
    int64_t tx = 0, ty = 0;
    for (int j = j0; j < std::min(count, j0 + KWY_BATCH_MAX); ++j) { tx = std::max(tx, jobs[j].x_length); ty = std::max(ty, jobs[j].y_length); }
    bytes += kwy_pad(dtw_layout(tx, ty, dim, radius, false, nullptr) * (size_t)std::min(KWY_BATCH_MAX, count - j0));
  }
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  for (int j0 = 0; j0 < count; j0 += KWY_BATCH_MAX) {
    const int n = std::min(KWY_BATCH_MAX, count - j0);
    dtw_pair pr[KWY_BATCH_MAX];
    int *status[KWY_BATCH_MAX];
    for (int j = 0; j < n; ++j) {
      const kwy_dtw_job &q = jobs[j0 + j];
      pr[j] = dtw_pair{q.x, q.y, (int)q.x_length, (int)q.y_length, q.dist, q.path, q.path_len};
    }
    KWY_TRY(fastdtw_core(ctx, pr, n, dim, radius, false, status));
  }
  return KWY_OK;
}

extern "C" int kwy_fastdtw_dev(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
                               int radius, double *dist, int32_t *path, int64_t *path_len) {
  KWY_TRY(dtw_check(ctx, x, Tx, y, Ty, dim, radius, dist, path, path_len));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, dtw_layout(Tx, Ty, dim, radius, false, nullptr)));
  int *status;
  const dtw_pair pr = {x, y, (int)Tx, (int)Ty, dist, path, path_len};
  return fastdtw_core(ctx, &pr, 1, dim, radius, false, &status);
}

extern "C" int kwy_fastdtw(kwy_ctx *ctx, const double *x, int64_t Tx, const double *y, int64_t Ty, int dim,
                           int radius, double *dist, int32_t *path, int64_t *path_len) {
  KWY_TRY(dtw_check(ctx, x, Tx, y, Ty, dim, radius, dist, path, path_len));
  KWY_HIP(hipSetDevice(ctx->device));
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool full = attempt == 1;     // (the capacities are bounds: the second attempt is a safeguard only)
    size_t bx = kwy_pad(sizeof(double) * Tx * dim), by = kwy_pad(sizeof(double) * Ty * dim);
    size_t bp = kwy_pad(sizeof(int32_t) * 2 * (Tx + Ty + 2));
    KWY_TRY(kwy_arena_begin(ctx, dtw_layout(Tx, Ty, dim, radius, full, nullptr) + bx + by + bp + 2 * kwy_pad(64)));
    double *dx = kwy_arena<double>(ctx, (size_t)Tx * dim), *dy = kwy_arena<double>(ctx, (size_t)Ty * dim);
    int32_t *dpath = kwy_arena<int32_t>(ctx, 2 * (Tx + Ty + 2));
    double *ddist = kwy_arena<double>(ctx, 8);
    int64_t *dlen = kwy_arena<int64_t>(ctx, 8);
    KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * Tx * dim, hipMemcpyHostToDevice, ctx->stream));
    KWY_HIP(hipMemcpyAsync(dy, y, sizeof(double) * Ty * dim, hipMemcpyHostToDevice, ctx->stream));
    int *status;
    const dtw_pair pr = {dx, dy, (int)Tx, (int)Ty, ddist, dpath, dlen};
    KWY_TRY(fastdtw_core(ctx, &pr, 1, dim, radius, full, &status));
    int hstatus = 0;
    int64_t hlen = 0;
    KWY_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipMemcpyAsync(&hlen, dlen, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipMemcpyAsync(dist, ddist, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    KWY_HIP(hipStreamSynchronize(ctx->stream));
    if (hstatus != 0) continue;  // band storage exceeded
    if (hlen > Tx + Ty) { ctx->err = "fastdtw: path overflow"; return KWY_EHIP; }
    KWY_HIP(hipMemcpy(path, dpath, sizeof(int32_t) * 2 * hlen, hipMemcpyDeviceToHost));
    *path_len = hlen;
    return KWY_OK;
  }
  ctx->err = "fastdtw: band storage overflow";
  return KWY_EHIP;
}
