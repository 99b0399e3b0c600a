// kwy_dio.hip -- DIO f0 estimation and StoneMask refinement on gfx950.
//
// Replaces pyworld.dio / pyworld.stonemask (reference call sites
// kwiiyatta/vocoder/world.py:35-40; WORLD dio.cpp / stonemask.cpp as shipped with
// pyworld 0.2.8).  SURVEY.md ranks this producer of the f0 track as 8(f)-1: it
// sits in front of the hot path, and is here so that wav-in analysis needs no CPU
// numerics and -- round 5 -- no trip over the host either: kwy_dio_batch_dev /
// kwy_stonemask_batch_dev take device pointers for up to 16 utterances per pass
// of launches and do not synchronise.
//
// DIO on the CPU filters the whole signal with FFTs of ~2^19 points: a 50 Hz
// low-cut, then one Nuttall low-pass per half-octave band.  Both are short FIRs
// (1921 and <= 956 taps at 48 kHz), so the filtered signal of band b is ONE
// linear convolution of the centred signal with h_b = lowcut * nuttall_b.  Round
// 1-4 evaluated the two stages as direct LDS-tiled convolutions (2.3 G
// multiply-adds per 10 s utterance: as long as the whole analysis); now a
// workgroup takes a block of 8192 samples, transforms it ONCE (real FFT in LDS,
// the spectrum stays in registers) and multiplies it with each band's
// precomputed filter spectrum: overlap-save, 8 transforms per 5 315 outputs of
// all 7 bands instead of 20 k multiply-adds per output.  The filtered signals
// never reach memory: the same workgroup finds the zero crossings of its block
// (four interval tracks per band) while the block is in LDS and leaves them as
// short lists, which a tiny kernel strings together in order.  Then per band
// the tracks' interpolation onto the frame times, the candidate / score per
// frame; finally the best-candidate contour and WORLD's four-step contour repair.
//
// StoneMask: one workgroup per frame; the two windowed spectra are only needed
// at <= 6 harmonic bins, so they are evaluated as direct DFT sums with the same
// twiddle table an FFT would use.
#include <math.h>

#include <complex>
#include <vector>

#include "kwy_internal.hpp"

#define DIO_SAFE 0.000000000001
#define DIO_MAXVAL 100000.0
#define DIO_CUTOFF 50.0
#define DIO_MAX_BANDS 16
#define DIO_LOG2H 12                 // the overlap-save transform: N = 8192 reals = 4096 packed complex
#define DIO_H (1 << DIO_LOG2H)
#define DIO_N (2 * DIO_H)
#define DIO_NT 512                   // threads of the filter kernel: one radix-8 butterfly per thread and pass
#define DIO_PARTS 256                // partial sums of the mean

// A pass takes up to 32 utterances (twice the other batch entries' 16: 40 bytes of descriptor each, by value): the
// contour repair is ONE wavefront per utterance walking ~2 000 dependent look-ups -- 0.49 ms however many utterances
// walk side by side, alone on the chip inside a step's stream -- so the 32 utterances of a wave of 16 pairs share it.
#define DIO_BATCH 32
// ---- one pass of launches over <= DIO_BATCH utterances -------------------------------------------
// Everything a kernel needs travels by value in this struct (no descriptor in device memory: capturable in a HIP
// graph).  Scratch: one block per utterance at scratch + u * stride, laid out for the LONGEST utterance of the pass.
struct dio_utt {
  const double *x;      // n samples
  double *tpos, *f0;    // T frame times / f0 values (written)
  int *status;          // one word: set to 1 when an engine's zero-crossing buffer overflowed
  int n, T;
};

struct dio_plan {
  int count, nbands;
  int ny_max, T_max, ntiles_max, cap_max;   // layout extents (ntiles_max: filter blocks of the longest utterance)
  int cap_block;                            // edges one filter block may leave per engine
  int half, V;                              // Lh + 2 max_hal (the longest combined filter is 2 half taps); outputs per block
  double fs, f0_floor, f0_ceil, frame_period, allowed_range;
  double boundary[DIO_MAX_BANDS];
  const kwy_c *G;                           // [nbands][DIO_H + 1]: filter spectra / N, band delays equalised
  char *scratch;
  int64_t stride;
  int64_t off_part, off_lists, off_cnt, off_nedges, off_fine, off_cand, off_score, off_w1, off_w2, off_idx, off_trans;
  dio_utt u[DIO_BATCH];
  template <class T>
  __device__ __forceinline__ T *at(int utt, int64_t off) const { return (T *)(scratch + utt * stride + off); }
};

// ---- zero-crossing engines -------------------------------------------------------------
// engine e = 4*band + kind; kind 0: f, 1: -f, 2: f[i+1]-f[i], 3: f[i]-f[i+1]
__device__ __forceinline__ double dio_sig(const double *__restrict__ f, int kind, int i) {
  switch (kind) {
    case 0: return f[i];
    case 1: return -f[i];
    case 2: return -f[i] - (-f[i + 1]);
    default: return -(-f[i] - (-f[i + 1]));
  }
}

__device__ __forceinline__ bool dio_is_edge(const double *__restrict__ f, int kind, int i, int len) {
  // negative-going point between samples i and i+1 of the engine's signal (edge index i+1)
  return i < len - 1 && 0.0 < dio_sig(f, kind, i) && dio_sig(f, kind, i + 1) <= 0.0;
}

// ---- signal preparation ---------------------------------------------------------------
// partial sums of an utterance's samples (grid: DIO_PARTS x count); the mean is their sum in index order / (n + 1):
// x[n] = 0 is part of the average, as upstream
__global__ __launch_bounds__(KWY_THREADS) void k_dio_sum(dio_plan P) {
  __shared__ double red[8];
  const int utt = blockIdx.y;
  const double *__restrict__ x = P.u[utt].x;
  const int n = P.u[utt].n;
  double s = 0.0;
  for (int i = blockIdx.x * KWY_THREADS + threadIdx.x; i < n; i += gridDim.x * KWY_THREADS) s += x[i];
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) {
    P.at<double>(utt, P.off_part)[blockIdx.x] = s;
    if (blockIdx.x == 0) *P.u[utt].status = 0;
  }
}

// Overlap-save block `blockIdx.x` of utterance `blockIdx.y`: outputs i0 .. i0 + V - 1 of every band.
//   y[m]       = x[m] - mean on [0, ny) (x[n] = 0), 0 elsewhere
//   filt_b[i]  = sum_c h_b[c] y[i + D_b - c],  h_b = lowcut * nuttall_b,  D_b = Lh + 2 hal_b
// (upstream: ylc = lowcut filter of y, then the band's Nuttall window over ylc; both "same"-centred).  The segment
// seg[q] = y[i0 - half + 1 + q], q < N, is transformed once; G_b carries h_b delayed by 2 (max_hal - hal_b), so that
// for every band output i sits at q = i - i0 + 2 half - 1, free of wrap-around for q >= 2 half - 1.
__global__ __launch_bounds__(DIO_NT, 4) void k_dio_filter(dio_plan P, const kwy_c *__restrict__ twH,
                                                         const kwy_c *__restrict__ twN) {
  constexpr int H = DIO_H, N = DIO_N, NT = DIO_NT;
  extern __shared__ double smem[];
  kwy_c *z = (kwy_c *)smem;              // H + 1 complex
  kwy_c *twl = z + (H + 1);              // exp(-2 pi i k / H), k < H/8
  double *sh = (double *)(twl + H / 8);  // DIO_PARTS + 8
  int (*zc)[128] = (int (*)[128])(sh + DIO_PARTS + 8);     // [4][128]: edges per (engine of the band, row, wavefront)
  const int tid = threadIdx.x, utt = blockIdx.y;
  const int n = P.u[utt].n, ny = n + 1;
  const int i0 = blockIdx.x * P.V;
  if (i0 >= ny) {       // (uniform) a block behind this utterance's end: no edges
    int *cnt0 = P.at<int>(utt, P.off_cnt);
    for (int e_ = tid; e_ < 4 * P.nbands; e_ += NT) cnt0[e_ * P.ntiles_max + blockIdx.x] = 0;
    return;
  }
  const double *__restrict__ x = P.u[utt].x;
  if (tid < DIO_PARTS) sh[tid] = P.at<double>(utt, P.off_part)[tid];
  for (int i = tid; i < H / 8; i += NT) twl[i] = twH[i];
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < DIO_PARTS; ++i) m += sh[i];
    sh[DIO_PARTS] = m / ny;
  }
  __syncthreads();
  const double mean = sh[DIO_PARTS];
  double *A = (double *)z;
  const int s0 = i0 - P.half + 1;
#pragma unroll
  for (int j = 0; j < N / NT; ++j) {
    const int q = tid + NT * j, m = s0 + q;
    double v = 0.0;
    if (m >= 0 && m < ny) v = (m < n ? x[m] : 0.0) - mean;
    A[q] = v;
  }
  __syncthreads();
  const kwy_c twb = twN[tid];
  kwy_rfft_inplace<DIO_LOG2H, NT>(z, twl, twb, twN);
  kwy_c X[H / NT];
#pragma unroll
  for (int r = 0; r < H / NT; ++r) X[r] = z[tid + NT * r];
  const double xh = z[H].x;
  __syncthreads();
  const int first = 2 * P.half - 1;
  const int wv = tid >> 6, lane = tid & 63;
  const int rows = (P.V + NT - 1) / NT;                    // <= DIO_ZROWS
  const unsigned long long below = (1ull << lane) - 1ull;
  int *cnt = P.at<int>(utt, P.off_cnt);
  for (int b = 0; b < P.nbands; ++b) {
    const kwy_c *__restrict__ G = P.G + (size_t)b * (H + 1);
#pragma unroll
    for (int r = 0; r < H / NT; ++r) z[tid + NT * r] = cmulf(G[tid + NT * r], X[r]);
    if (tid == 0) z[H] = {G[H].x * xh, 0.0};
    kwy_irfft_inplace<DIO_LOG2H, NT>(z, twl, twb, twN);
    // ---- the block's zero crossings, straight from LDS: fl[li] = filtered sample i0 + li, li < V + 2 (the block
    //      OWNS li < V; the two samples behind them are its own values too, so every edge is decided exactly once,
    //      on one set of numbers).  Pass 1 counts per (engine, row, wavefront) by ballots, a scan orders the slots,
    //      pass 2 evaluates the same tests again and writes the edges to the block's list.
    const double *fl = A + first;
    // the four engines' tests on one triple of samples: f, -f, and the differences d(li) = f[li+1] - f[li] (formed as
    // the engines' own expression -f[li] - (-f[li+1])), d(li+1)
    auto edges = [&](int li, bool (&e)[4]) {
      const int i = i0 + li;
      const bool own = li < P.V;
      const double f0_ = fl[min(li, P.V + 1)], f1_ = fl[min(li + 1, P.V + 1)], f2_ = fl[min(li + 2, P.V + 1)];
      const double d0 = -f0_ - (-f1_), d1 = -f1_ - (-f2_);
      e[0] = own && i < ny - 1 && 0.0 < f0_ && f1_ <= 0.0;
      e[1] = own && i < ny - 1 && 0.0 < -f0_ && -f1_ <= 0.0;
      e[2] = own && i < ny - 2 && 0.0 < d0 && d1 <= 0.0;
      e[3] = own && i < ny - 2 && 0.0 < -d0 && -d1 <= 0.0;
    };
    unsigned int rows_with_edges = 0;      // (uniform per wavefront) rows in which this wavefront found any edge
    for (int j = 0; j < rows; ++j) {
      bool e[4];
      edges(tid + NT * j, e);
      unsigned long long any = 0;
#pragma unroll
      for (int kind = 0; kind < 4; ++kind) {
        const unsigned long long m = __ballot(e[kind]);
        any |= m;
        if (lane == 0) zc[kind][j * (NT / 64) + wv] = __popcll(m);
      }
      if (any) rows_with_edges |= 1u << j;
    }
    __syncthreads();
    if (wv < 4) {                        // wavefront k: exclusive scan of engine k's <= 128 slots, two per lane
      const int nslot = rows * (NT / 64);
      const int v0 = 2 * lane < nslot ? zc[wv][2 * lane] : 0, v1 = 2 * lane + 1 < nslot ? zc[wv][2 * lane + 1] : 0;
      int inc = v0 + v1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
      }
      zc[wv][2 * lane] = inc - v0 - v1;
      zc[wv][2 * lane + 1] = inc - v1;
      if (lane == 63) cnt[(4 * b + wv) * P.ntiles_max + blockIdx.x] = min(inc, P.cap_block);
    }
    __syncthreads();
    double *lists = P.at<double>(utt, P.off_lists);
    for (int j = 0; j < rows; ++j) {
      if (!((rows_with_edges >> j) & 1u)) continue;
      const int li = tid + NT * j;
      bool e[4];
      edges(li, e);
#pragma unroll
      for (int kind = 0; kind < 4; ++kind) {
        const unsigned long long m = __ballot(e[kind]);
        if (e[kind]) {
          const int at = zc[kind][j * (NT / 64) + wv] + __popcll(m & below);
          if (at < P.cap_block) {
            const double a = dio_sig(fl, kind, li), c = dio_sig(fl, kind, li + 1);
            lists[((int64_t)(4 * b + kind) * P.ntiles_max + blockIdx.x) * P.cap_block + at] = (i0 + li + 1) - a / (c - a);
          } else {
            atomicExch(P.u[utt].status, 1);
          }
        }
      }
    }
    __syncthreads();
  }
}

// exclusive scan of every engine's per-block edge counts (one workgroup per engine and utterance)
__global__ __launch_bounds__(KWY_THREADS) void k_dio_zc_scan(dio_plan P) {
  __shared__ int tot[KWY_THREADS];
  const int utt = blockIdx.y, ntiles = P.ntiles_max;
  int *c = P.at<int>(utt, P.off_cnt) + blockIdx.x * ntiles;
  const int t = threadIdx.x;
  const int chunk = (ntiles + KWY_THREADS - 1) / KWY_THREADS;
  const int b0 = t * chunk, b1 = min(ntiles, b0 + chunk);
  int run = 0;
  for (int i = b0; i < b1; ++i) run += c[i];
  tot[t] = run;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < KWY_THREADS; ++i) { int v = tot[i]; tot[i] = acc; acc += v; }
    P.at<int>(utt, P.off_nedges)[blockIdx.x] = acc;
  }
  __syncthreads();
  run = tot[t];
  for (int i = b0; i < b1; ++i) { int v = c[i]; c[i] = run; run += v; }
}

// the blocks' lists strung together in order: fine[e][off .. off + c) <- list of (e, block); grid (blocks, engines, utterances)
__global__ __launch_bounds__(64) void k_dio_zc_gather(dio_plan P) {
  const int blk = blockIdx.x, e = blockIdx.y, utt = blockIdx.z;
  const int ny = P.u[utt].n + 1, cap = ny / 8 + 64;
  const int *cnt = P.at<int>(utt, P.off_cnt) + e * P.ntiles_max;
  const int off = cnt[blk];
  const int end = blk + 1 < P.ntiles_max ? cnt[blk + 1] : P.at<int>(utt, P.off_nedges)[e];
  const int c = end - off;
  if (c <= 0) return;
  const double *src = P.at<double>(utt, P.off_lists) + ((int64_t)e * P.ntiles_max + blk) * P.cap_block;
  double *o = P.at<double>(utt, P.off_fine) + (int64_t)e * P.cap_max;
  for (int k = threadIdx.x; k < c; k += 64) {
    if (off + k < cap) o[off + k] = src[k];
    else atomicExch(P.u[utt].status, 1);
  }
}

// ---- candidates per (band, frame) -----------------------------------------------------------
// WORLD interp1 (histc bucketing, linear extrapolation through the end segments) of the
// interval track k of an engine, evaluated at xi.  The track has n = nedges-1 points:
// location[k] = (fe[k]+fe[k+1])/2/fs, interval[k] = fs/(fe[k+1]-fe[k]).
__device__ inline double dio_interp(const double *__restrict__ fe, int n, double fs, double xi) {
  auto loc = [&](int k) { return (fe[k] + fe[k + 1]) / 2.0 / fs; };
  auto itv = [&](int k) { return fs / (fe[k + 1] - fe[k]); };
  // number of locations <= xi
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (loc(mid) <= xi) lo = mid + 1; else hi = mid; }
  int k = lo;
  if (k < 1) k = 1;
  if (k > n - 1) k = n - 1;
  const double xa = loc(k - 1), xb = loc(k);
  const double s = (xi - xa) / (xb - xa);
  const double ya = itv(k - 1), yb = itv(k);
  return ya + s * (yb - ya);
}

// grid: (frames of the longest utterance / 256, bands, utterances)
__global__ void k_dio_candidates(dio_plan P) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y, utt = blockIdx.z;
  const int T = P.u[utt].T;
  if (j >= T) return;
  const int cap = (P.u[utt].n + 1) / 8 + 64;
  const int *ne = P.at<int>(utt, P.off_nedges) + 4 * b;
  const double *fine = P.at<double>(utt, P.off_fine);
  double c = 0.0, sc = DIO_MAXVAL;
  // every engine needs at least 3 interval points (count - 2 > 0)
  if (ne[0] - 1 > 2 && ne[1] - 1 > 2 && ne[2] - 1 > 2 && ne[3] - 1 > 2) {
    const double xi = j * P.frame_period / 1000.0;
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = dio_interp(fine + (int64_t)(4 * b + k) * P.cap_max, min(ne[k], cap) - 1, P.fs, xi);
    c = (v[0] + v[1] + v[2] + v[3]) / 4.0;
    sc = sqrt(((v[0] - c) * (v[0] - c) + (v[1] - c) * (v[1] - c) + (v[2] - c) * (v[2] - c) +
               (v[3] - c) * (v[3] - c)) / 3.0);
    const double bf = P.boundary[b];
    if (c > bf || c < bf / 2.0 || c > P.f0_ceil || c < P.f0_floor) { c = 0.0; sc = DIO_MAXVAL; }
  }
  P.at<double>(utt, P.off_cand)[(int64_t)b * P.T_max + j] = c;
  P.at<double>(utt, P.off_score)[(int64_t)b * P.T_max + j] = sc / (c + DIO_SAFE);
}

// ---- best contour + WORLD FixF0Contour -----------------------------------------------------------
// SelectBestF0 of WORLD's contour repair: of the frame's candidates the one closest to the linear prediction
// (3 current - past) / 2 (the first of equals), or none if it is further than allowed_range from it.  Returns the
// candidate's band, 0xff for none.
#define DIO_NONE 0xff
__host__ __device__ static inline int dio_trans_row(int nb) { return (nb * (nb + 1) + 7) & ~7; }

template <class CAND>
__device__ __forceinline__ int dio_select_best(double current_f0, double past_f0, CAND cand, int nb, double allowed_range) {
  const double reference_f0 = (current_f0 * 3.0 - past_f0) / 2.0;
  double minimum_error = fabs(reference_f0 - cand(0));
  double best = cand(0);
  int at = 0;
  for (int i = 1; i < nb; ++i) {
    const double c = cand(i), err = fabs(reference_f0 - c);
    if (err < minimum_error) { minimum_error = err; best = c; at = i; }
  }
  if (fabs(1.0 - best / reference_f0) > allowed_range) return DIO_NONE;
  return at;
}

// The repair's steps 3 and 4 grow the voiced sections frame by frame, each new value chosen from the NEXT frame's
// candidates given the two values before it -- which are themselves candidates of their frames (or 0).  So the walk
// moves through a finite state space, (band of the current value, band of the value before it | none), and its
// transitions can all be evaluated beforehand, in parallel: trans[dir][frame][a (nb + 1) + b] = the band chosen in
// frame + 1 (dir 0) / frame - 1 (dir 1) when the current value is candidate a of `frame` and the one before it
// candidate b of the frame behind (b = nb: 0).  The serial walk (k_dio_fix) is then one table look-up per frame
// instead of seven comparisons and a division on its dependence chain (0.8 ms per 16 utterances before).
// grid: (states of the longest utterance / 256, 2, utterances)
__global__ __launch_bounds__(KWY_THREADS) void k_dio_trans(dio_plan p) {
  const int utt = blockIdx.z, dir = blockIdx.y, nb = p.nbands, T = p.u[utt].T, Ts = p.T_max;
  const int RS = dio_trans_row(nb), per = nb * (nb + 1);
  const int64_t g = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x;
  const int frame = (int)(g / per), st = (int)(g - (int64_t)frame * per);
  if (frame >= T) return;
  const int a = st / (nb + 1), b = st - a * (nb + 1);
  const double *__restrict__ cand = p.at<double>(utt, p.off_cand);
  const int behind = dir == 0 ? frame - 1 : frame + 1, ahead = dir == 0 ? frame + 1 : frame - 1;
  int res = DIO_NONE;
  if (ahead >= 0 && ahead < T) {
    const double cur = cand[(int64_t)a * Ts + frame];
    const double past = (b < nb && behind >= 0 && behind < T) ? cand[(int64_t)b * Ts + behind] : 0.0;
    res = dio_select_best(cur, past, [&](int i) { return cand[(int64_t)i * Ts + ahead]; }, nb, p.allowed_range);
    // (a chosen candidate that IS 0 ends the walk exactly like "none": folded here, so that the walk's dependence
    // chain is the table look-up alone)
    if (res != DIO_NONE && cand[(int64_t)res * Ts + ahead] == 0.0) res = DIO_NONE;
  }
  p.at<unsigned char>(utt, p.off_trans)[((int64_t)dir * Ts + frame) * RS + st] = (unsigned char)res;
}

// one workgroup per utterance
__global__ __launch_bounds__(KWY_THREADS) void k_dio_fix(dio_plan p) {
  const int utt = blockIdx.x;
  const int T = p.u[utt].T, Ts = p.T_max, nb = p.nbands, tid = threadIdx.x;
  const double *__restrict__ cand = p.at<double>(utt, p.off_cand);
  const double *__restrict__ score = p.at<double>(utt, p.off_score);
  double *__restrict__ w1 = p.at<double>(utt, p.off_w1), *__restrict__ w2 = p.at<double>(utt, p.off_w2);
  int *__restrict__ idxbuf = p.at<int>(utt, p.off_idx);
  double *__restrict__ tpos = p.u[utt].tpos, *__restrict__ f0 = p.u[utt].f0;
  for (int i = tid; i < T; i += KWY_THREADS) {
    tpos[i] = i * p.frame_period / 1000.0;
    f0[i] = 0.0;
    double tmp = score[i], best = cand[i];
    for (int j = 1; j < nb; ++j)
      if (tmp > score[(int64_t)j * Ts + i]) { tmp = score[(int64_t)j * Ts + i]; best = cand[(int64_t)j * Ts + i]; }
    w1[i] = best;  // best_f0_contour
  }
  __syncthreads();
  const int vrm = (int)(0.5 + 1000.0 / p.frame_period / p.f0_floor) * 2 + 1;
  if (T <= vrm) return;
  // step 1 -> w2
  for (int i = tid; i < T; i += KWY_THREADS) {
    auto base = [&](int k) { return (k < vrm || k >= T - vrm) ? 0.0 : w1[k]; };
    double v = 0.0;
    if (i >= vrm) {
      double fb = base(i);
      v = fabs((fb - base(i - 1)) / (DIO_SAFE + fb)) < p.allowed_range ? fb : 0.0;
    }
    w2[i] = v;
  }
  __syncthreads();
  // step 2 -> f0 (used as scratch f0_step2)
  const int center = (vrm - 1) / 2;
  for (int i = tid; i < T; i += KWY_THREADS) {
    double v = w2[i];
    if (i >= center && i < T - center)
      for (int j = -center; j <= center; ++j)
        if (w2[i + j] == 0) { v = 0.0; break; }
    f0[i] = v;
  }
  __syncthreads();
  // ---- the voiced sections' borders, in frame order: every thread looks at a stretch of frames, an exclusive scan
  //      of the two counts places its finds (round 1-4: one thread walked all T frames through global memory, 1.5 ms)
  __shared__ int sh2[2][KWY_WAVES];
  __shared__ int s_tot[2];
  int *positive_index = idxbuf, *negative_index = idxbuf + T;
  {
    const int chunk = (T - 1 + KWY_THREADS - 1) / KWY_THREADS;
    const int i0 = 1 + tid * chunk, i1 = min(T, i0 + chunk);
    int np_ = 0, nn_ = 0;
    for (int i = i0; i < i1; ++i) {
      const double a = f0[i - 1], b = f0[i];
      if (b == 0 && a != 0) ++nn_;
      else if (a == 0 && b != 0) ++np_;
    }
    const int lane = tid & 63, wv = tid >> 6;
    int ip = np_, in_ = nn_;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(ip, o), un = __shfl_up(in_, o);
      if (lane >= o) { ip += up; in_ += un; }
    }
    if (lane == 63) { sh2[0][wv] = ip; sh2[1][wv] = in_; }
    __syncthreads();
    int op = ip - np_, on = in_ - nn_;
    for (int w = 0; w < wv; ++w) { op += sh2[0][w]; on += sh2[1][w]; }
    if (tid == KWY_THREADS - 1) { s_tot[0] = op + np_; s_tot[1] = on + nn_; }
    for (int i = i0; i < i1; ++i) {
      const double a = f0[i - 1], b = f0[i];
      if (b == 0 && a != 0) negative_index[on++] = i - 1;
      else if (a == 0 && b != 0) positive_index[op++] = i;
    }
  }
  __syncthreads();
  // ---- steps 3 and 4: the sections grow along the candidates, serially.  ONE wavefront walks, all lanes in step
  //      (every lane loads, decides and stores the same values: a lane always sees its own earlier stores), through
  //      the transition tables of k_dio_trans: per frame one look-up in LDS, where the tables and the candidates of
  //      64 frames at a time are staged (coalesced loads).
  if (tid < 64) {
    __shared__ double cc[DIO_MAX_BANDS][64];
    __shared__ unsigned int tr[64 * (DIO_MAX_BANDS * (DIO_MAX_BANDS + 1) + 7) / 4 + 8];
    const int lane = tid, RS = dio_trans_row(nb), nbp = nb + 1;
    const unsigned char *__restrict__ trans = p.at<unsigned char>(utt, p.off_trans);
    const int pc = s_tot[0], nc = s_tot[1];
    for (int dir = 0; dir < 2; ++dir) {
      int c0 = -64;
      auto stage = [&](int idx) {            // frames [idx & ~63, +64): candidates and this direction's transitions
        c0 = idx & ~63;
        for (int b = 0; b < nb; ++b) cc[b][lane] = (c0 + lane < T) ? cand[(int64_t)b * Ts + c0 + lane] : 0.0;
        const unsigned int *src = (const unsigned int *)(trans + ((int64_t)dir * Ts + c0) * RS);   // RS % 8 == 0
        const int words = min(64, T - c0) * RS / 4;
        for (int q = lane; q < words; q += 64) tr[q] = src[q];
      };
      const int sections = dir == 0 ? nc : pc;
      const int *__restrict__ list = dir == 0 ? negative_index : positive_index;
      for (int k = 0; k < sections; ++k) {
        // step 3 (dir 0): sections in frame order, forward from their last frame up to the next section's last;
        // step 4 (dir 1): sections from the last to the first, backward from their first frame
        const int i = dir == 0 ? k : pc - 1 - k;
        const int step = dir == 0 ? 1 : -1;
        // this section's border and the next one's in one go (two lanes, one round trip)
        const int nb_i = dir == 0 ? (i == nc - 1 ? -1 : i + 1) : (i == 0 ? -1 : i - 1);
        const int mine = lane == 0 ? list[i] : (lane == 1 && nb_i >= 0 ? list[nb_i] : 0);
        const int j0 = __builtin_amdgcn_readlane(mine, 0), jn = __builtin_amdgcn_readlane(mine, 1);
        const int limit = dir == 0 ? (nb_i < 0 ? T - 1 : jn) : (nb_i < 0 ? 1 : jn);
        // the start state: which bands' candidates ARE the two values the walk starts from.  Everything it takes is
        // loaded at once (lanes 0..15: the candidates of frame j0, lanes 16..31: of the frame behind, every lane:
        // the two values), so a section costs one memory round trip, not one per comparison.
        const int fb = j0 - step;
        const double v0 = f0[j0], v1 = f0[fb];
        double cv = 0.0;
        if (lane < nb) cv = cand[(int64_t)lane * Ts + j0];
        else if (lane >= 16 && lane < 16 + nb) cv = cand[(int64_t)(lane - 16) * Ts + fb];
        const unsigned long long ma = __ballot(lane < nb && cv == v0);
        const unsigned long long mb = __ballot(lane >= 16 && lane < 16 + nb && cv == v1) >> 16;
        int a = ma ? __builtin_ctzll(ma) : nb;
        int b = (v1 != 0.0 && mb) ? __builtin_ctzll(mb) : nb;
        for (int j = j0; dir == 0 ? j < limit : j > limit; j += step) {
          const int nx = j + step;
          if ((j & ~63) != c0) stage(j);
          int n = DIO_NONE;
          if (a < nb) n = ((const unsigned char *)tr)[(j - c0) * RS + a * nbp + b];
          if (n == DIO_NONE) { f0[nx] = 0.0; break; }
          if ((nx & ~63) != c0) stage(nx);      // (the new value is a candidate of frame nx)
          f0[nx] = cc[n][nx - c0];
          b = a;
          a = n;
        }
      }
    }
  }
}

// ---- StoneMask ------------------------------------------------------------------------------------
struct sm_view {
  const double *x, *tpos, *f0in;
  double *out;
  int x_length;
};
typedef kwy_batch<sm_view> sm_batch;
struct sm_tables { const kwy_c *t[20]; };   // t[l]: exp(-2 pi i k / 2^l)

// one workgroup per frame of every utterance of the batch
__global__ __launch_bounds__(KWY_THREADS) void k_stonemask(sm_batch batch, int fs, sm_tables tw_tables) {
  __shared__ double part[KWY_WAVES][24];
  __shared__ double res[32];
  const int tid = threadIdx.x;
  const int utt = batch.find(blockIdx.x);
  const int64_t frame = (int)blockIdx.x - batch.start[utt];
  const double *__restrict__ x = batch.u[utt].x;
  const double *__restrict__ tpos = batch.u[utt].tpos;
  double *__restrict__ out = batch.u[utt].out;
  const int x_length = batch.u[utt].x_length;
  const double initial_f0 = batch.u[utt].f0in[frame];
  if (initial_f0 <= 40.0 || initial_f0 > fs / 12.0) {
    if (tid == 0) out[frame] = 0.0;
    return;
  }
  const double pos = tpos[frame];
  const int half = (int)(1.5 * fs / initial_f0 + 1.0);
  const double wlt = (2.0 * half + 1.0) / fs;
  const int len = 2 * half + 1;
  const int log2n = 2 + (int)(log(half * 2.0 + 1.0) / 0.69314718055994529);
  const int N = 1 << log2n;
  const kwy_c *tw = tw_tables.t[log2n];
  const double base_time0 = (double)(-half) / fs;
  const int basic_index = kwy_matlab_round((pos + base_time0) * fs + 0.001);
  // WORLD's window: mainw(i) = 0.42 + 0.5 cos(th_i) + 0.08 cos(2 th_i), th_i = 2 pi ((basic_index + i - 1) / fs - pos)
  // / wlt, and its difference window -(mainw(i+1) - mainw(i-1)) / 2.  One sincos per sample serves both: the
  // neighbours' angles differ by the constant step dth, so with c+- = cos(th +- dth)
  //     mainw(i+1) - mainw(i-1) = 0.5 (c+ - c-) + 0.16 (c+^2 - c-^2) = -sin th sin dth (1 + 0.64 cos th cos dth)
  // (rounds 1-4 evaluated six cosines per sample and pass: 0.98 ms per 32 utterances, as much as CheapTrick.)
  const double fsd = (double)fs;
  double sd, cd;
  sincos(2.0 * KWY_PI / (fsd * wlt), &sd, &cd);
  auto mainw_of = [](double c) { return 0.42 + 0.5 * c + 0.08 * (2.0 * c * c - 1.0); };
  double f0_try = initial_f0;
  double mean_f0 = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int nh = pass == 0 ? 2 : min((int)(fsd / 2.0 / initial_f0), 6);
    int bins[6];
#pragma unroll
    for (int h = 0; h < 6; ++h)
      bins[h] = __builtin_amdgcn_readfirstlane(min(kwy_matlab_round(f0_try * N / fsd * (h + 1)), N / 2));
    // the bins' twiddles exp(-2 pi i bin i / N) at i = tid + 256 r: the first from the table, the others by rotation
    // with the (uniform) factor of 256 samples
    kwy_c wh[6], ws[6];
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      wh[h] = ws[h] = kwy_c{1.0, 0.0};
      if (h < nh) {
        wh[h] = tw[(int)(((int64_t)bins[h] * tid) & (N - 1))];
        ws[h] = tw[(int)(((int64_t)bins[h] * KWY_THREADS) & (N - 1))];
      }
    }
    // accumulate main/diff spectra at the nh bins
    double acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.0;
    for (int i = tid; i < len; i += KWY_THREADS) {
      const int idx = max(0, min(x_length - 1, basic_index + i - 1));
      const double xv = x[idx];
      double st, ct;
      sincos(2.0 * KWY_PI * ((basic_index + i - 1.0) / fs - pos) / wlt, &st, &ct);
      const double mw = mainw_of(ct);
      double dw;
      if (i == 0) dw = -mainw_of(ct * cd - st * sd) / 2.0;
      else if (i == len - 1) dw = mainw_of(ct * cd + st * sd) / 2.0;
      else dw = st * sd * (1.0 + 0.64 * ct * cd) / 2.0;
      const double a = xv * mw, b = xv * dw;
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        if (h < nh) {
          const kwy_c w = wh[h];
          acc[4 * h + 0] += a * w.x; acc[4 * h + 1] += a * w.y;
          acc[4 * h + 2] += b * w.x; acc[4 * h + 3] += b * w.y;
          wh[h] = cmulf(w, ws[h]);
        }
      }
    }
    // all sums with ONE workgroup barrier: every wavefront reduces its 4 nh values by lane shifts and leaves them in
    // LDS, then the four partial sums are added in wave order (rounds 1-4: 24 block reductions of two barriers each)
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      if (q < 4 * nh) {
        const double v = kwy_wave_sum(acc[q]);
        if ((tid & 63) == 0) part[tid >> 6][q] = v;
      }
    }
    __syncthreads();
    if (tid < 4 * nh) res[tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    __syncthreads();
    // FixF0 (all threads compute the same scalars)
    double numerator = 0.0, denominator = 0.0;
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      if (h < nh) {
        const double mr = res[4 * h], mi = res[4 * h + 1], dr = res[4 * h + 2], di = res[4 * h + 3];
        const double power = mr * mr + mi * mi;
        const double num_i = mr * di - mi * dr;
        const double inst = power == 0.0 ? 0.0 : (double)bins[h] * fsd / N + num_i / power * fsd / 2.0 / KWY_PI;
        const double amp = sqrt(power);
        numerator += amp * inst;
        denominator += amp * (h + 1);
      }
    }
    const double est = numerator / (denominator + DIO_SAFE);
    __syncthreads();
    if (pass == 0) {
      if (est <= 0.0 || est > initial_f0 * 2) { mean_f0 = 0.0; break; }
      f0_try = est;
    } else {
      mean_f0 = est;
    }
  }
  if (fabs(mean_f0 - initial_f0) > initial_f0 * 0.2) mean_f0 = initial_f0;
  if (tid == 0) out[frame] = mean_f0;
}

// ---- host side ---------------------------------------------------------------------------------------
static void nuttall_host(int n, double *y) {
  for (int i = 0; i < n; ++i) {
    double tmp = i / (n - 1.0);
    y[i] = 0.355768 - 0.487396 * cos(2.0 * KWY_PI * tmp) + 0.144232 * cos(4.0 * KWY_PI * tmp) -
           0.012604 * cos(6.0 * KWY_PI * tmp);
  }
}

static inline int mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }

// in-place radix-2 FFT of N = 2^k points on the host (the filter spectra are built once per parameter set)
static void host_fft(std::vector<std::complex<double>> &a) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  std::vector<std::complex<double>> w(n / 2);
  for (size_t k = 0; k < n / 2; ++k) {
    const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
    w[k] = {(double)cosl(ang), (double)sinl(ang)};
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t step = n / len;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w[k * step];
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
  }
}

struct dio_filters {
  int nbands, Lh, max_hal;
  double boundary[DIO_MAX_BANDS];
  const kwy_c *G;
};

// The band filters' spectra for (fs, f0_floor, f0_ceil, channels_in_octave), cached per context:
// G_b[k] = DFT_N(lowcut)[k] * DFT_N(nuttall_b delayed by 2 (max_hal - hal_b))[k] / N,  k <= N/2
static int dio_get_filters(kwy_ctx *ctx, int fs, double f0_floor, double f0_ceil, double channels_in_octave,
                           dio_filters *out) {
  dio_filters f;
  f.nbands = 1 + (int)(log(f0_ceil / f0_floor) / 0.69314718055994529 * channels_in_octave);
  if (f.nbands > DIO_MAX_BANDS) { ctx->err = "dio: too many bands"; return KWY_EINVAL; }
  for (int i = 0; i < f.nbands; ++i) f.boundary[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);
  f.Lh = mround((double)fs / DIO_CUTOFF);
  const int Nlc = 2 * f.Lh + 1;
  std::vector<int> hal(f.nbands);
  f.max_hal = 0;
  for (int b = 0; b < f.nbands; ++b) {
    hal[b] = mround(fs / f.boundary[b] / 2.0);
    if (hal[b] < 1) { ctx->err = "dio: band too high for this sampling rate"; return KWY_EINVAL; }
    f.max_hal = hal[b] > f.max_hal ? hal[b] : f.max_hal;
  }
  if (2 * (f.Lh + 2 * f.max_hal) > DIO_N - 1024) {
    ctx->err = "dio: filters too long for this sampling rate / f0 floor (the overlap-save block is 8192 samples)";
    return KWY_EINVAL;
  }
  char key[160];
  snprintf(key, sizeof(key), "dio:%d:%.17g:%.17g:%.17g", fs, f0_floor, f0_ceil, channels_in_octave);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    const int N = DIO_N, H = DIO_H;
    std::vector<std::complex<double>> lc(N, 0.0), nu(N);
    {
      // DesignLowCutFilter: -(Hanning)/sum, centre tap + 1
      std::vector<double> g(Nlc);
      double sum = 0.0;
      for (int i = 1; i <= Nlc; ++i) { g[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * KWY_PI / (Nlc + 1)); sum += g[i - 1]; }
      for (int i = 0; i < Nlc; ++i) g[i] = -g[i] / sum;
      g[f.Lh] += 1.0;
      for (int i = 0; i < Nlc; ++i) lc[i] = g[i];
    }
    host_fft(lc);
    std::vector<kwy_c> G((size_t)f.nbands * (H + 1));
    std::vector<double> nt;
    for (int b = 0; b < f.nbands; ++b) {
      nt.assign(4 * hal[b], 0.0);
      nuttall_host(4 * hal[b], nt.data());
      std::fill(nu.begin(), nu.end(), std::complex<double>(0.0, 0.0));
      const int delay = 2 * (f.max_hal - hal[b]);
      for (int j = 0; j < 4 * hal[b]; ++j) nu[delay + j] = nt[j];
      host_fft(nu);
      for (int k = 0; k <= H; ++k) {
        const std::complex<double> v = lc[k] * nu[k] / (double)N;
        G[(size_t)b * (H + 1) + k] = {v.real(), v.imag()};
      }
    }
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(kwy_c) * G.size()));
    KWY_HIP(hipMemcpy(d, G.data(), sizeof(kwy_c) * G.size(), hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  f.G = (const kwy_c *)it->second;
  *out = f;
  return KWY_OK;
}

// lays out one utterance block for the extents in p (ny_max, T_max, ...) and returns its size
static size_t dio_layout(dio_plan &p) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += kwy_pad(bytes); return (int64_t)o; };
  const int nengines = 4 * p.nbands;
  p.off_part = take(sizeof(double) * DIO_PARTS);
  p.off_lists = take(sizeof(double) * (size_t)nengines * p.ntiles_max * p.cap_block);
  p.off_cnt = take(sizeof(int) * (size_t)nengines * p.ntiles_max);
  p.off_nedges = take(sizeof(int) * nengines);
  p.off_fine = take(sizeof(double) * (size_t)nengines * p.cap_max);
  p.off_cand = take(sizeof(double) * (size_t)p.nbands * p.T_max);
  p.off_score = take(sizeof(double) * (size_t)p.nbands * p.T_max);
  p.off_w1 = take(sizeof(double) * p.T_max);
  p.off_w2 = take(sizeof(double) * p.T_max);
  p.off_idx = take(sizeof(int) * 2 * (size_t)p.T_max);
  p.off_trans = take(2 * (size_t)p.T_max * dio_trans_row(p.nbands));
  p.stride = (int64_t)off;
  return off;
}

// parameter checks, the cached filter spectra, and the plan (layout for utterances of up to n_max samples)
static int dio_prepare(kwy_ctx *ctx, int fs, double f0_floor, double f0_ceil, double channels_in_octave,
                       double frame_period_ms, int speed, double allowed_range, int64_t n_max, dio_plan *out,
                       size_t *block) {
  if (fs <= 0 || !(f0_floor > 0) || !(f0_ceil > f0_floor) || !(channels_in_octave > 0) || !(frame_period_ms > 0) ||
      n_max <= 0 || n_max > 0x3fffffff) {
    ctx->err = "dio: bad argument";
    return KWY_EINVAL;
  }
  if (speed != 1) { ctx->err = "dio: only speed=1 (pyworld's default, the value kwiiyatta uses) is implemented"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  dio_filters f;
  KWY_TRY(dio_get_filters(ctx, fs, f0_floor, f0_ceil, channels_in_octave, &f));
  dio_plan p;
  p.count = 0;
  p.nbands = f.nbands;
  for (int i = 0; i < DIO_MAX_BANDS; ++i) p.boundary[i] = i < f.nbands ? f.boundary[i] : 0.0;
  p.G = f.G;
  p.half = f.Lh + 2 * f.max_hal;
  p.V = DIO_N - 2 * p.half + 1 - 2;     // outputs a block OWNS: two more are valid behind them (the edge tests' look-ahead)
  p.fs = fs; p.f0_floor = f0_floor; p.f0_ceil = f0_ceil; p.frame_period = frame_period_ms;
  p.allowed_range = allowed_range;
  p.ny_max = (int)n_max + 1;
  p.T_max = (int)kwy_dio_frames(fs, n_max, frame_period_ms);
  p.ntiles_max = (p.ny_max + p.V - 1) / p.V;
  p.cap_block = p.V / 8 + 64;
  p.cap_max = p.ny_max / 8 + 64;
  p.scratch = nullptr;
  *block = dio_layout(p);
  *out = p;
  return KWY_OK;
}

// One pass of launches over jobs[0 .. count), count <= DIO_BATCH (device pointers); scratch: `count` blocks of
// the plan's stride; a job without a status word gets spare[u].
static int dio_pass(kwy_ctx *ctx, const kwy_f0_job *jobs, int count, dio_plan p, char *scratch, int *spare) {
  p.count = count;
  p.scratch = scratch;
  for (int u = 0; u < DIO_BATCH; ++u) {
    if (u < count) {
      const kwy_f0_job &q = jobs[u];
      p.u[u] = dio_utt{q.x, q.temporal_positions, q.f0, q.status ? (int *)q.status : spare + u, (int)q.x_length,
                       (int)kwy_dio_frames((int)p.fs, q.x_length, p.frame_period)};
    } else {
      p.u[u] = dio_utt{nullptr, nullptr, nullptr, nullptr, 0, 0};
    }
  }
  const kwy_c *twH, *twN;
  KWY_TRY(kwy_get_twiddles(ctx, DIO_LOG2H, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, DIO_LOG2H + 1, &twN));
  const size_t lds = sizeof(kwy_c) * (DIO_H + 1 + DIO_H / 8) + sizeof(double) * (DIO_PARTS + 8) + sizeof(int) * 4 * 128;
  KWY_HIP(hipFuncSetAttribute((const void *)k_dio_filter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nengines = 4 * p.nbands;
  const unsigned nblocks = (unsigned)p.ntiles_max;
  hipLaunchKernelGGL(k_dio_sum, dim3(DIO_PARTS, count), dim3(KWY_THREADS), 0, ctx->stream, p);
  KWY_PROF(ctx, "k_dio_filter", hipLaunchKernelGGL(k_dio_filter, dim3(nblocks, count), dim3(DIO_NT), lds, ctx->stream,
                                                   p, twH, twN));
  {
    kwy_prof_scope ps_(ctx, "k_dio_zc");
    hipLaunchKernelGGL(k_dio_zc_scan, dim3(nengines, count), dim3(KWY_THREADS), 0, ctx->stream, p);
    hipLaunchKernelGGL(k_dio_zc_gather, dim3(nblocks, nengines, count), dim3(64), 0, ctx->stream, p);
  }
  KWY_PROF(ctx, "k_dio_candidates", hipLaunchKernelGGL(k_dio_candidates, dim3((p.T_max + 255) / 256, p.nbands, count),
                                                       dim3(256), 0, ctx->stream, p));
  {
    kwy_prof_scope ps_(ctx, "k_dio_fix");
    const int64_t states = (int64_t)p.T_max * p.nbands * (p.nbands + 1);
    hipLaunchKernelGGL(k_dio_trans, dim3((unsigned)((states + KWY_THREADS - 1) / KWY_THREADS), 2, count), dim3(KWY_THREADS),
                       0, ctx->stream, p);
    hipLaunchKernelGGL(k_dio_fix, dim3(count), dim3(KWY_THREADS), 0, ctx->stream, p);
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// pyworld.dio for `count` utterances (device pointers, not synchronised): passes of <= DIO_BATCH utterances
extern "C" int kwy_dio_batch_dev(kwy_ctx *ctx, const kwy_f0_job *jobs, int count, int fs, double f0_floor,
                                 double f0_ceil, double channels_in_octave, double frame_period_ms, int speed,
                                 double allowed_range) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 1) { ctx->err = "dio: bad argument"; return KWY_EINVAL; }
  int64_t n_max = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_f0_job &q = jobs[i];
    if (!q.x || !q.temporal_positions || !q.f0 || q.x_length <= 0 || q.x_length > 0x3fffffff) {
      ctx->err = "dio: bad argument";
      return KWY_EINVAL;
    }
    n_max = q.x_length > n_max ? q.x_length : n_max;
  }
  dio_plan p;
  size_t block;
  KWY_TRY(dio_prepare(ctx, fs, f0_floor, f0_ceil, channels_in_octave, frame_period_ms, speed, allowed_range, n_max,
                      &p, &block));
  const int per_pass = count < DIO_BATCH ? count : DIO_BATCH;
  KWY_TRY(kwy_arena_begin(ctx, block * per_pass + kwy_pad(sizeof(int) * DIO_BATCH)));
  char *scratch = (char *)kwy_arena_alloc(ctx, block * per_pass);
  int *spare = kwy_arena<int>(ctx, DIO_BATCH);
  if (!scratch || !spare) { ctx->err = "dio: scratch arena too small"; return KWY_ENOMEM; }
  // (the passes of a call share the scratch: they run one after the other on the context's stream)
  for (int i0 = 0; i0 < count; i0 += DIO_BATCH)
    KWY_TRY(dio_pass(ctx, jobs + i0, count - i0 < DIO_BATCH ? count - i0 : DIO_BATCH, p, scratch, spare));
  return KWY_OK;
}

extern "C" int kwy_dio_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, double f0_floor, double f0_ceil,
                           double channels_in_octave, double frame_period_ms, int speed, double allowed_range,
                           double *temporal_positions, double *f0, int32_t *status) {
  const kwy_f0_job job = {x, x_length, temporal_positions, f0, status};
  return kwy_dio_batch_dev(ctx, &job, 1, fs, f0_floor, f0_ceil, channels_in_octave, frame_period_ms, speed,
                           allowed_range);
}

// host pointers: the same pass on staged copies (a batch of one), synchronous
extern "C" int kwy_dio(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, double f0_floor,
                       double f0_ceil, double channels_in_octave, double frame_period_ms, int speed,
                       double allowed_range, double *temporal_positions, double *f0) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !temporal_positions || !f0) { ctx->err = "dio: bad argument"; return KWY_EINVAL; }
  dio_plan p;
  size_t block;
  KWY_TRY(dio_prepare(ctx, fs, f0_floor, f0_ceil, channels_in_octave, frame_period_ms, speed, allowed_range, x_length,
                      &p, &block));
  const int64_t T = p.T_max;
  KWY_TRY(kwy_arena_begin(ctx, block + kwy_pad(sizeof(double) * x_length) + 2 * kwy_pad(sizeof(double) * T) +
                                   kwy_pad(sizeof(int) * KWY_BATCH_MAX)));
  char *scratch = (char *)kwy_arena_alloc(ctx, block);
  double *dx = kwy_arena<double>(ctx, x_length), *dt = kwy_arena<double>(ctx, T), *df0 = kwy_arena<double>(ctx, T);
  int *dstatus = kwy_arena<int>(ctx, KWY_BATCH_MAX);
  if (!scratch || !dx || !dt || !df0 || !dstatus) { ctx->err = "dio: scratch arena too small"; return KWY_ENOMEM; }
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * x_length, hipMemcpyHostToDevice, ctx->stream));
  const kwy_f0_job job = {dx, x_length, dt, df0, nullptr};
  KWY_TRY(dio_pass(ctx, &job, 1, p, scratch, dstatus));
  int hstatus = 0;
  KWY_HIP(hipMemcpyAsync(temporal_positions, dt, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(f0, df0, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(&hstatus, dstatus, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  if (hstatus != 0) { ctx->err = "dio: zero-crossing buffer overflow (signal too noisy for the band filters)"; return KWY_EHIP; }
  return KWY_OK;
}

// ---- StoneMask entries ---------------------------------------------------------------------------------
static int sm_tables_for(kwy_ctx *ctx, int fs, sm_tables *tabs) {
  // twiddle tables for every FFT size a frame may ask for (f0 in (40, fs/12])
  const int max_half = (int)(1.5 * fs / 40.0 + 1.0);
  const int max_log2 = 2 + (int)(log(max_half * 2.0 + 1.0) / 0.69314718055994529);
  if (max_log2 >= 20) { ctx->err = "stonemask: sampling rate too high"; return KWY_EINVAL; }
  for (int l = 0; l < 20; ++l) tabs->t[l] = nullptr;
  for (int l = 2; l <= max_log2; ++l) KWY_TRY(kwy_get_twiddles(ctx, l, &tabs->t[l]));
  return KWY_OK;
}

static int sm_launch(kwy_ctx *ctx, const kwy_utterance *utts, int count, int fs, const sm_tables &tabs) {
  for (int i0 = 0; i0 < count; i0 += KWY_BATCH_MAX) {
    sm_batch b;
    b.n = count - i0 < KWY_BATCH_MAX ? count - i0 : KWY_BATCH_MAX;
    b.start[0] = 0;
    for (int u = 0; u < b.n; ++u) {
      const kwy_utterance &q = utts[i0 + u];
      b.u[u] = sm_view{q.x, q.temporal_positions, q.f0, q.out, (int)q.x_length};
      b.start[u + 1] = b.start[u] + (int)q.f0_length;
    }
    KWY_PROF(ctx, "k_stonemask", hipLaunchKernelGGL(k_stonemask, dim3((unsigned)b.start[b.n]), dim3(KWY_THREADS), 0,
                                                    ctx->stream, b, fs, tabs));
    KWY_HIP(hipGetLastError());
  }
  return KWY_OK;
}

// pyworld.stonemask for `count` utterances (device pointers; kwy_utterance.out = the refined f0, f0_length values);
// one grid over all frames of <= KWY_BATCH_MAX utterances, not synchronised
extern "C" int kwy_stonemask_batch_dev(kwy_ctx *ctx, const kwy_utterance *utts, int count, int fs) {
  if (!ctx) return KWY_EINVAL;
  if (!utts || count < 1 || fs <= 0) { ctx->err = "stonemask: bad argument"; return KWY_EINVAL; }
  int64_t frames = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_utterance &q = utts[i];
    if (!q.x || !q.temporal_positions || !q.f0 || !q.out || q.x_length <= 0 || q.x_length > 0x7fffffff ||
        q.f0_length <= 0) {
      ctx->err = "stonemask: bad argument";
      return KWY_EINVAL;
    }
    frames += q.f0_length;
  }
  if (frames > 0x7fffffff) { ctx->err = "stonemask: too many frames"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  sm_tables tabs;
  KWY_TRY(sm_tables_for(ctx, fs, &tabs));
  return sm_launch(ctx, utts, count, fs, tabs);
}

extern "C" int kwy_stonemask_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                                 const double *f0, int64_t T, double *refined_f0) {
  const kwy_utterance u = {x, x_length, t, f0, T, refined_f0};
  return kwy_stonemask_batch_dev(ctx, &u, 1, fs);
}

extern "C" int kwy_stonemask(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                             const double *f0, int64_t T, double *refined_f0) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !t || !f0 || !refined_f0 || x_length <= 0 || x_length > 0x7fffffff || T <= 0 || fs <= 0) {
    ctx->err = "stonemask: bad argument";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  sm_tables tabs;
  KWY_TRY(sm_tables_for(ctx, fs, &tabs));
  size_t bx = kwy_pad(sizeof(double) * x_length), bt = kwy_pad(sizeof(double) * T);
  KWY_TRY(kwy_arena_begin(ctx, bx + 3 * bt));
  double *dx = kwy_arena<double>(ctx, x_length), *dt = kwy_arena<double>(ctx, T);
  double *df0 = kwy_arena<double>(ctx, T), *dout = kwy_arena<double>(ctx, T);
  if (!dx || !dt || !df0 || !dout) { ctx->err = "stonemask: scratch arena too small"; return KWY_ENOMEM; }
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * x_length, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dt, t, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(df0, f0, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  const kwy_utterance u = {dx, x_length, dt, df0, T, dout};
  KWY_TRY(sm_launch(ctx, &u, 1, fs, tabs));
  KWY_HIP(hipMemcpyAsync(refined_f0, dout, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
