// kwy_dio.hip -- DIO f0 estimation and StoneMask refinement on gfx950.
//
// Replaces pyworld.dio / pyworld.stonemask (reference call sites
// kwiiyatta/vocoder/world.py:35-40; WORLD dio.cpp / stonemask.cpp as shipped with
// pyworld 0.2.8).  SURVEY.md ranks this producer of the f0 track as 8(f)-1: it
// sits in front of the timed hot path, and is here so that wav-in analysis
// needs no CPU numerics at all.
//
// DIO on the CPU filters the whole signal with FFTs of ~2^19 points (a 50 Hz
// low-cut, then one Nuttall low-pass per half-octave band).  All filters are
// short FIRs (<= 1921 taps), so here they are direct, LDS-tiled convolutions --
// embarrassingly parallel and free of the big-FFT round trips.  Then, per band:
// the four zero-crossing interval tracks (ordered stream compaction), their
// interpolation onto the frame times, the candidate / score per frame; finally
// the best-candidate contour and WORLD's four-step contour repair.
//
// StoneMask: one workgroup per frame; the two windowed spectra are only needed
// at <= 6 harmonic bins, so they are evaluated as direct DFT sums with the same
// twiddle table an FFT would use.
#include <math.h>

#include <vector>

#include "kwy_internal.hpp"

#define DIO_SAFE 0.000000000001
#define DIO_MAXVAL 100000.0
#define DIO_CUTOFF 50.0
#define DIO_MAX_BANDS 16
#define DIO_CONV_OUT 1024   // outputs per convolution workgroup (4 per thread)

// ---- signal preparation ---------------------------------------------------------------
__global__ __launch_bounds__(KWY_THREADS) void k_dio_sum(const double *__restrict__ x, int n,
                                                        double *__restrict__ partial) {
  __shared__ double red[8];
  double s = 0.0;
  for (int i = blockIdx.x * KWY_THREADS + threadIdx.x; i < n; i += gridDim.x * KWY_THREADS) s += x[i];
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// y[m] = x[m] - mean on [0, ny) (x[n] = 0 is part of the average, as upstream), 0 elsewhere;
// stored with an offset of E zeros on both sides.
__global__ void k_dio_center(const double *__restrict__ x, int n, int ny, int E,
                             const double *__restrict__ partial, int nparts, double *__restrict__ y) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x - E;
  if (m >= ny + E) return;
  double mean = 0.0;
  for (int i = 0; i < nparts; ++i) mean += partial[i];
  mean /= ny;
  double v = 0.0;
  if (m >= 0 && m < ny) v = (m < n ? x[m] : 0.0) - mean;
  y[m + E] = v;
}

// out[m] = sum_{k=0}^{ntaps-1} taps[k] * in[m + shift - k]   for m in [m0, m1)
// `in` and `out` are stored with offsets in_off / out_off; reads outside
// [in_lo, in_hi) return 0.
struct conv_desc {
  int ntaps, shift, m0, m1, in_lo, in_hi, in_off, out_off;
  int64_t taps_off, out_stride;
};

__global__ __launch_bounds__(KWY_THREADS) void k_dio_conv(const double *__restrict__ in,
                                                         const double *__restrict__ taps_all,
                                                         const conv_desc *__restrict__ descs,
                                                         double *__restrict__ out_all) {
  extern __shared__ double smem[];
  const conv_desc d = descs[blockIdx.y];
  const int base = d.m0 + blockIdx.x * DIO_CONV_OUT;
  if (base >= d.m1) return;
  double *tp = smem;               // ntaps
  double *seg = smem + d.ntaps;    // DIO_CONV_OUT + ntaps - 1 input samples
  const double *taps = taps_all + d.taps_off;
  for (int k = threadIdx.x; k < d.ntaps; k += KWY_THREADS) tp[k] = taps[k];
  // seg[q] = in[base + shift - (ntaps-1) + q]
  const int seg_n = DIO_CONV_OUT + d.ntaps - 1;
  const int first = base + d.shift - (d.ntaps - 1);
  for (int q = threadIdx.x; q < seg_n; q += KWY_THREADS) {
    int idx = first + q;
    seg[q] = (idx >= d.in_lo && idx < d.in_hi) ? in[idx + d.in_off] : 0.0;
  }
  __syncthreads();
  double *out = out_all + d.out_stride * blockIdx.y;
  // thread t computes outputs base + t + 256*r, r < 4
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int t = threadIdx.x;
  for (int k = 0; k < d.ntaps; ++k) {
    const double c = tp[k];
    const int q = t + (d.ntaps - 1) - k;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += c * seg[q + KWY_THREADS * r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int m = base + t + KWY_THREADS * r;
    if (m < d.m1) out[m + d.out_off] = acc[r];
  }
}

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

#define DIO_ZC_PER_THREAD 8
#define DIO_ZC_TILE (KWY_THREADS * DIO_ZC_PER_THREAD)

__device__ __forceinline__ int dio_block_exscan(int v, int *sh, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  __syncthreads();
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < KWY_WAVES; ++i) { if (i < wv) woff += sh[i]; tot += sh[i]; }
  *total = tot;
  return woff + (inc - v);
}

__global__ __launch_bounds__(KWY_THREADS) void k_dio_zc_count(const double *__restrict__ filt,
                                                             int64_t fstride, int ny, int ntiles,
                                                             int *__restrict__ cnt) {
  __shared__ int sh[KWY_WAVES];
  const int e = blockIdx.y, band = e >> 2, kind = e & 3;
  const double *f = filt + fstride * band;
  const int len = kind < 2 ? ny : ny - 1;
  const int base = blockIdx.x * DIO_ZC_TILE + threadIdx.x * DIO_ZC_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < DIO_ZC_PER_THREAD; ++j) c += dio_is_edge(f, kind, base + j, len) ? 1 : 0;
  int tot;
  (void)dio_block_exscan(c, sh, &tot);
  if (threadIdx.x == 0) cnt[e * ntiles + blockIdx.x] = tot;
}

// exclusive scan of every engine's tile counts (one block per engine)
__global__ __launch_bounds__(KWY_THREADS) void k_dio_zc_scan(int *__restrict__ cnt, int ntiles,
                                                            int *__restrict__ nedges) {
  __shared__ int tot[KWY_THREADS];
  int *c = cnt + blockIdx.x * ntiles;
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
    nedges[blockIdx.x] = acc;
  }
  __syncthreads();
  run = tot[t];
  for (int i = b0; i < b1; ++i) { int v = c[i]; c[i] = run; run += v; }
}

__global__ __launch_bounds__(KWY_THREADS) void k_dio_zc_emit(const double *__restrict__ filt,
                                                            int64_t fstride, int ny, int ntiles,
                                                            const int *__restrict__ cnt, int cap,
                                                            double *__restrict__ fine, int *__restrict__ status) {
  __shared__ int sh[KWY_WAVES];
  const int e = blockIdx.y, band = e >> 2, kind = e & 3;
  const double *f = filt + fstride * band;
  const int len = kind < 2 ? ny : ny - 1;
  const int base = blockIdx.x * DIO_ZC_TILE + threadIdx.x * DIO_ZC_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < DIO_ZC_PER_THREAD; ++j) c += dio_is_edge(f, kind, base + j, len) ? 1 : 0;
  int tot;
  int pos = cnt[e * ntiles + blockIdx.x] + dio_block_exscan(c, sh, &tot);
  double *o = fine + (int64_t)e * cap;
#pragma unroll
  for (int j = 0; j < DIO_ZC_PER_THREAD; ++j) {
    const int i = base + j;
    if (dio_is_edge(f, kind, i, len)) {
      if (pos < cap) {
        const double a = dio_sig(f, kind, i), b = dio_sig(f, kind, i + 1);
        o[pos] = (i + 1) - a / (b - a);
      } else {
        atomicExch(status, 1);
      }
      ++pos;
    }
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

struct dio_params {
  int ny, T, nbands, cap;
  double fs, f0_floor, f0_ceil, frame_period, allowed_range;
  double boundary[DIO_MAX_BANDS];
};

__global__ void k_dio_candidates(const double *__restrict__ fine, const int *__restrict__ nedges,
                                 dio_params p, double *__restrict__ cand, double *__restrict__ score) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= p.T) return;
  const int *ne = nedges + 4 * b;
  double c = 0.0, sc = DIO_MAXVAL;
  // every engine needs at least 3 interval points (count - 2 > 0)
  if (ne[0] - 1 > 2 && ne[1] - 1 > 2 && ne[2] - 1 > 2 && ne[3] - 1 > 2) {
    const double xi = j * p.frame_period / 1000.0;
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = dio_interp(fine + (int64_t)(4 * b + k) * p.cap, min(ne[k], p.cap) - 1, p.fs, xi);
    c = (v[0] + v[1] + v[2] + v[3]) / 4.0;
    sc = sqrt(((v[0] - c) * (v[0] - c) + (v[1] - c) * (v[1] - c) + (v[2] - c) * (v[2] - c) +
               (v[3] - c) * (v[3] - c)) / 3.0);
    const double bf = p.boundary[b];
    if (c > bf || c < bf / 2.0 || c > p.f0_ceil || c < p.f0_floor) { c = 0.0; sc = DIO_MAXVAL; }
  }
  cand[(int64_t)b * p.T + j] = c;
  score[(int64_t)b * p.T + j] = sc / (c + DIO_SAFE);
}

// ---- best contour + WORLD FixF0Contour (one workgroup) -----------------------------------------
__device__ inline double dio_select_best(double current_f0, double past_f0, const double *__restrict__ cand,
                                         int nb, int T, int idx, double allowed_range) {
  const double reference_f0 = (current_f0 * 3.0 - past_f0) / 2.0;
  double minimum_error = fabs(reference_f0 - cand[idx]);
  double best = cand[idx];
  for (int i = 1; i < nb; ++i) {
    double err = fabs(reference_f0 - cand[(int64_t)i * T + idx]);
    if (err < minimum_error) { minimum_error = err; best = cand[(int64_t)i * T + idx]; }
  }
  if (fabs(1.0 - best / reference_f0) > allowed_range) return 0.0;
  return best;
}

__global__ __launch_bounds__(KWY_THREADS) void k_dio_fix(const double *__restrict__ cand,
                                                        const double *__restrict__ score, dio_params p,
                                                        double *__restrict__ w1, double *__restrict__ w2,
                                                        int *__restrict__ idxbuf, double *__restrict__ tpos,
                                                        double *__restrict__ f0) {
  const int T = p.T, nb = p.nbands, tid = threadIdx.x;
  __shared__ int s_pos, s_neg;
  for (int i = tid; i < T; i += KWY_THREADS) {
    tpos[i] = i * p.frame_period / 1000.0;
    f0[i] = 0.0;
    double tmp = score[i], best = cand[i];
    for (int j = 1; j < nb; ++j)
      if (tmp > score[(int64_t)j * T + i]) { tmp = score[(int64_t)j * T + i]; best = cand[(int64_t)j * T + i]; }
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
  // sections + steps 3 and 4: short, serial
  if (tid == 0) {
    int *positive_index = idxbuf, *negative_index = idxbuf + T;
    int pc = 0, nc = 0;
    for (int i = 1; i < T; ++i) {
      if (f0[i] == 0 && f0[i - 1] != 0) negative_index[nc++] = i - 1;
      else if (f0[i - 1] == 0 && f0[i] != 0) positive_index[pc++] = i;
    }
    // step 3 (forward) in place on f0
    for (int i = 0; i < nc; ++i) {
      int limit = i == nc - 1 ? T - 1 : negative_index[i + 1];
      for (int j = negative_index[i]; j < limit; ++j) {
        f0[j + 1] = dio_select_best(f0[j], f0[j - 1], cand, nb, T, j + 1, p.allowed_range);
        if (f0[j + 1] == 0) break;
      }
    }
    // step 4 (backward)
    for (int i = pc - 1; i >= 0; --i) {
      int limit = i == 0 ? 1 : positive_index[i - 1];
      for (int j = positive_index[i]; j > limit; --j) {
        f0[j - 1] = dio_select_best(f0[j], f0[j + 1], cand, nb, T, j - 1, p.allowed_range);
        if (f0[j - 1] == 0) break;
      }
    }
    s_pos = pc; s_neg = nc;
  }
}

// ---- StoneMask ------------------------------------------------------------------------------------
__global__ __launch_bounds__(KWY_THREADS) void k_stonemask(const double *__restrict__ x, int x_length, int fs,
                                                          const double *__restrict__ tpos,
                                                          const double *__restrict__ f0in,
                                                          const kwy_c *const *__restrict__ tw_tables,
                                                          double *__restrict__ out) {
  __shared__ double red[8];
  __shared__ double res[32];
  const int tid = threadIdx.x;
  const int64_t frame = blockIdx.x;
  const double initial_f0 = f0in[frame];
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
  const kwy_c *tw = tw_tables[log2n];
  const double base_time0 = (double)(-half) / fs;
  const int basic_index = kwy_matlab_round((pos + base_time0) * fs + 0.001);
  auto mainw = [&](int i) {
    double tmp = (basic_index + i - 1.0) / fs - pos;
    return 0.42 + 0.5 * cos(2.0 * KWY_PI * tmp / wlt) + 0.08 * cos(4.0 * KWY_PI * tmp / wlt);
  };
  const double fsd = (double)fs;
  double f0_try = initial_f0;
  double mean_f0 = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int nh = pass == 0 ? 2 : min((int)(fsd / 2.0 / initial_f0), 6);
    int bins[6];
#pragma unroll
    for (int h = 0; h < 6; ++h) bins[h] = min(kwy_matlab_round(f0_try * N / fsd * (h + 1)), N / 2);
    // accumulate main/diff spectra at the nh bins
    double acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.0;
    for (int i = tid; i < len; i += KWY_THREADS) {
      const int idx = max(0, min(x_length - 1, basic_index + i - 1));
      const double xv = x[idx];
      const double mw = mainw(i);
      double dw;
      if (i == 0) dw = -mainw(1) / 2.0;
      else if (i == len - 1) dw = mainw(len - 2) / 2.0;
      else dw = -(mainw(i + 1) - mainw(i - 1)) / 2.0;
      const double a = xv * mw, b = xv * dw;
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        if (h < nh) {
          const kwy_c w = tw[(int)(((int64_t)bins[h] * i) & (N - 1))];
          acc[4 * h + 0] += a * w.x; acc[4 * h + 1] += a * w.y;
          acc[4 * h + 2] += b * w.x; acc[4 * h + 3] += b * w.y;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      if (q < 4 * nh) {
        double v = kwy_block_sum(acc[q], red);
        if (tid == 0) res[q] = v;
      }
    }
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

extern "C" int kwy_dio(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, double f0_floor,
                       double f0_ceil, double channels_in_octave, double frame_period_ms, int speed,
                       double allowed_range, double *temporal_positions, double *f0) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !temporal_positions || !f0 || x_length <= 0 || x_length > 0x3fffffff || fs <= 0 ||
      !(f0_floor > 0) || !(f0_ceil > f0_floor) || !(channels_in_octave > 0) || !(frame_period_ms > 0)) {
    ctx->err = "dio: bad argument";
    return KWY_EINVAL;
  }
  if (speed != 1) { ctx->err = "dio: only speed=1 (pyworld's default, the value kwiiyatta uses) is implemented"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const int n = (int)x_length, ny = n + 1;
  const int T = (int)kwy_dio_frames(fs, x_length, frame_period_ms);
  dio_params p;
  p.nbands = 1 + (int)(log(f0_ceil / f0_floor) / 0.69314718055994529 * channels_in_octave);
  if (p.nbands > DIO_MAX_BANDS) { ctx->err = "dio: too many bands"; return KWY_EINVAL; }
  for (int i = 0; i < p.nbands; ++i) p.boundary[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);
  p.ny = ny; p.T = T; p.fs = fs; p.f0_floor = f0_floor; p.f0_ceil = f0_ceil;
  p.frame_period = frame_period_ms; p.allowed_range = allowed_range;

  // filters: low-cut (centred, 2*Lh+1 taps) and one Nuttall low-pass per band
  const int Lh = mround((double)fs / DIO_CUTOFF);
  const int Nlc = 2 * Lh + 1;
  std::vector<int> hal(p.nbands);
  int max_hal = 0;
  size_t ntaps_total = Nlc;
  for (int b = 0; b < p.nbands; ++b) {
    hal[b] = mround(fs / p.boundary[b] / 2.0);
    if (hal[b] < 1) { ctx->err = "dio: band too high for this sampling rate"; return KWY_EINVAL; }
    max_hal = hal[b] > max_hal ? hal[b] : max_hal;
    ntaps_total += 4 * hal[b];
  }
  const int E = (Lh > 2 * max_hal ? Lh : 2 * max_hal) + 8;  // zero margin kept around y / ylc
  std::vector<double> taps(ntaps_total);
  {
    // DesignLowCutFilter: -(Hanning)/sum, centre tap + 1
    double sum = 0.0;
    for (int i = 1; i <= Nlc; ++i) { taps[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * KWY_PI / (Nlc + 1)); sum += taps[i - 1]; }
    for (int i = 0; i < Nlc; ++i) taps[i] = -taps[i] / sum;
    taps[Lh] += 1.0;
  }
  std::vector<conv_desc> descs(1 + p.nbands);
  const int64_t ylen = (int64_t)ny + 2 * E;
  // desc 0: ylc[m] = sum_k g[k-Lh] y[m - (k - Lh)], m in [-E, ny+E)
  descs[0] = {Nlc, Lh, -E, ny + E, -E, ny + E, E, E, 0, 0};
  size_t toff = Nlc;
  for (int b = 0; b < p.nbands; ++b) {
    nuttall_host(4 * hal[b], taps.data() + toff);
    // filtered[i] = sum_j nutt[j] * ylc[i + 2 hal - j], i in [0, ny)
    descs[1 + b] = {4 * hal[b], 2 * hal[b], 0, ny, -E, ny + E, E, 0, (int64_t)toff, (int64_t)ny};
    toff += 4 * hal[b];
  }
  const int ntiles = (ny + DIO_ZC_TILE - 1) / DIO_ZC_TILE;
  const int nengines = 4 * p.nbands;
  const int cap = ny / 8 + 64;
  p.cap = cap;

  size_t need = kwy_pad(sizeof(double) * n) + 2 * kwy_pad(sizeof(double) * ylen) + kwy_pad(sizeof(double) * 1024) +
                kwy_pad(sizeof(double) * ntaps_total) + kwy_pad(sizeof(conv_desc) * descs.size()) +
                kwy_pad(sizeof(double) * (size_t)ny * p.nbands) + kwy_pad(sizeof(int) * (size_t)nengines * ntiles) +
                kwy_pad(sizeof(int) * nengines) + kwy_pad(sizeof(double) * (size_t)nengines * cap) +
                2 * kwy_pad(sizeof(double) * (size_t)p.nbands * T) + 4 * kwy_pad(sizeof(double) * T) +
                kwy_pad(sizeof(int) * 2 * T) + kwy_pad(64);
  KWY_TRY(kwy_arena_begin(ctx, need));
  double *dx = kwy_arena<double>(ctx, n);
  double *dy = kwy_arena<double>(ctx, ylen), *dylc = kwy_arena<double>(ctx, ylen);
  double *dpart = kwy_arena<double>(ctx, 1024);
  double *dtaps = kwy_arena<double>(ctx, ntaps_total);
  conv_desc *ddesc = (conv_desc *)kwy_arena_alloc(ctx, sizeof(conv_desc) * descs.size());
  double *dfilt = kwy_arena<double>(ctx, (size_t)ny * p.nbands);
  int *dcnt = kwy_arena<int>(ctx, (size_t)nengines * ntiles);
  int *dnedges = kwy_arena<int>(ctx, nengines);
  double *dfine = kwy_arena<double>(ctx, (size_t)nengines * cap);
  double *dcand = kwy_arena<double>(ctx, (size_t)p.nbands * T), *dscore = kwy_arena<double>(ctx, (size_t)p.nbands * T);
  double *dw1 = kwy_arena<double>(ctx, T), *dw2 = kwy_arena<double>(ctx, T);
  double *dt = kwy_arena<double>(ctx, T), *df0 = kwy_arena<double>(ctx, T);
  int *didx = kwy_arena<int>(ctx, 2 * T);
  int *dstatus = kwy_arena<int>(ctx, 16);
  if (!dx || !dy || !dylc || !dpart || !dtaps || !ddesc || !dfilt || !dcnt || !dnedges || !dfine || !dcand ||
      !dscore || !dw1 || !dw2 || !dt || !df0 || !didx || !dstatus) {
    ctx->err = "dio: scratch arena too small";
    return KWY_ENOMEM;
  }
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dtaps, taps.data(), sizeof(double) * ntaps_total, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(ddesc, descs.data(), sizeof(conv_desc) * descs.size(), hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemsetAsync(dstatus, 0, sizeof(int) * 16, ctx->stream));

  const int nparts = 256;
  hipLaunchKernelGGL(k_dio_sum, dim3(nparts), dim3(KWY_THREADS), 0, ctx->stream, dx, n, dpart);
  hipLaunchKernelGGL(k_dio_center, dim3((unsigned)((ylen + 255) / 256)), dim3(256), 0, ctx->stream, dx, n, ny, E,
                     dpart, nparts, dy);
  {
    size_t lds = sizeof(double) * (Nlc + DIO_CONV_OUT + Nlc);
    // 96 kHz: 3841 low-cut taps -> 70 KB; the CU has 160 KB
    KWY_HIP(hipFuncSetAttribute((const void *)k_dio_conv, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    if (lds > 152 * 1024) { ctx->err = "dio: low-cut filter too long for this sampling rate"; return KWY_EINVAL; }
    hipLaunchKernelGGL(k_dio_conv, dim3((unsigned)((ylen + DIO_CONV_OUT - 1) / DIO_CONV_OUT), 1), dim3(KWY_THREADS),
                       lds, ctx->stream, dy, dtaps, ddesc, dylc);
    size_t lds2 = sizeof(double) * (4 * max_hal + DIO_CONV_OUT + 4 * max_hal);
    if (lds2 > 64 * 1024) { ctx->err = "dio: band filter too long for this sampling rate"; return KWY_EINVAL; }
    hipLaunchKernelGGL(k_dio_conv, dim3((unsigned)((ny + DIO_CONV_OUT - 1) / DIO_CONV_OUT), p.nbands),
                       dim3(KWY_THREADS), lds2, ctx->stream, dylc, dtaps, ddesc + 1, dfilt);
  }
  hipLaunchKernelGGL(k_dio_zc_count, dim3(ntiles, nengines), dim3(KWY_THREADS), 0, ctx->stream, dfilt, (int64_t)ny,
                     ny, ntiles, dcnt);
  hipLaunchKernelGGL(k_dio_zc_scan, dim3(nengines), dim3(KWY_THREADS), 0, ctx->stream, dcnt, ntiles, dnedges);
  hipLaunchKernelGGL(k_dio_zc_emit, dim3(ntiles, nengines), dim3(KWY_THREADS), 0, ctx->stream, dfilt, (int64_t)ny,
                     ny, ntiles, dcnt, cap, dfine, dstatus);
  hipLaunchKernelGGL(k_dio_candidates, dim3((T + 255) / 256, p.nbands), dim3(256), 0, ctx->stream, dfine, dnedges,
                     p, dcand, dscore);
  hipLaunchKernelGGL(k_dio_fix, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, dcand, dscore, p, dw1, dw2, didx, dt,
                     df0);
  KWY_HIP(hipGetLastError());
  int hstatus = 0;
  KWY_HIP(hipMemcpyAsync(temporal_positions, dt, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(f0, df0, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(&hstatus, dstatus, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  if (hstatus != 0) { ctx->err = "dio: zero-crossing buffer overflow (signal too noisy for the band filters)"; return KWY_EHIP; }
  return KWY_OK;
}

extern "C" int kwy_stonemask(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                             const double *f0, int64_t T, double *refined_f0) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !t || !f0 || !refined_f0 || x_length <= 0 || x_length > 0x7fffffff || T <= 0 || fs <= 0) {
    ctx->err = "stonemask: bad argument";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  // twiddle tables for every FFT size a frame may ask for (f0 in (40, fs/12])
  const int max_half = (int)(1.5 * fs / 40.0 + 1.0);
  const int max_log2 = 2 + (int)(log(max_half * 2.0 + 1.0) / 0.69314718055994529);
  if (max_log2 >= 20) { ctx->err = "stonemask: sampling rate too high"; return KWY_EINVAL; }
  std::vector<const kwy_c *> tabs(20, nullptr);
  for (int l = 2; l <= max_log2; ++l) KWY_TRY(kwy_get_twiddles(ctx, l, &tabs[l]));
  size_t bx = kwy_pad(sizeof(double) * x_length), bt = kwy_pad(sizeof(double) * T);
  KWY_TRY(kwy_arena_begin(ctx, bx + 3 * bt + kwy_pad(sizeof(void *) * 20)));
  double *dx = kwy_arena<double>(ctx, x_length), *dt = kwy_arena<double>(ctx, T);
  double *df0 = kwy_arena<double>(ctx, T), *dout = kwy_arena<double>(ctx, T);
  const kwy_c **dtabs = (const kwy_c **)kwy_arena_alloc(ctx, sizeof(void *) * 20);
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * x_length, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dt, t, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(df0, f0, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dtabs, tabs.data(), sizeof(void *) * 20, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_stonemask, dim3((unsigned)T), dim3(KWY_THREADS), 0, ctx->stream, dx, (int)x_length, fs, dt,
                     df0, dtabs, dout);
  KWY_HIP(hipGetLastError());
  KWY_HIP(hipMemcpyAsync(refined_f0, dout, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
