// kwy_d4c.hip -- D4C band aperiodicity (with the "LoveTrain" voicing gate) on gfx950.
//
// Replaces pyworld.d4c (reference call site kwiiyatta/vocoder/world.py:55;
// algorithm: Morise 2016, as shipped with pyworld 0.2.8).  Two frame-parallel
// kernels, one 256-thread workgroup per frame, everything staged in LDS:
//
//   k_d4c_lovetrain : Blackman-windowed frame -> power spectrum -> ratio of the
//                     cumulative power at 4 kHz / 7.9 kHz  (voicing gate)
//   k_d4c_body      : for gated frames: 2 temporal centroids (2 FFTs each),
//                     smoothed power spectrum, static group delay, then per
//                     3 kHz band a Nuttall-windowed FFT, an in-LDS sort and the
//                     energy ratio; finally the band values are interpolated to
//                     the K output bins.  Ungated frames write 1 - 1e-12.
//
// Algorithmic HBM bytes per frame: hop*8 + 16 in, K*8 out.
#include <math.h>

#include <vector>

#include "kwy_internal.hpp"

#define D4C_SAFE 0.000000000001
#define D4C_FLOOR_F0 47.0
#define D4C_FREQ_INTERVAL 3000.0
#define D4C_UPPER_LIMIT 15000.0
#define D4C_MAX_BANDS 8

enum { D4C_HANNING = 1, D4C_BLACKMAN = 2 };

// WORLD's F0-adaptive windows sampled at i = tid + NT r: value = g(cos(theta_i)), theta_i = pi cf0 (2 (i - half) /
// ratio) / fs.  theta advances by a constant from one r to the next, so only the first element evaluates the
// trigonometric polynomials; the others rotate (cos, sin) by the step angle (kwy_rotate: 6 instructions instead of
// ~40, error growth ~1 ulp per step over at most 16 steps -- far below the FFT rounding that follows).
__device__ __forceinline__ double d4c_window(int type, int i, int half, double ratio, int fs, double cf0) {
  const double c1 = kwy_cos_pi_range(KWY_PI * ((2.0 * (i - half) / ratio) / fs) * cf0);
  return type == D4C_HANNING ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);
}

struct d4c_window_walk {
  double c, s, cd, sd;
  __device__ __forceinline__ d4c_window_walk(int i0, int step, int half, double ratio, int fs, double cf0) {
    kwy_sincos_pi_range(KWY_PI * ((2.0 * (i0 - half) / ratio) / fs) * cf0, &s, &c);   // |theta| <= pi inside the window
    sincos(KWY_PI * ((2.0 * step / ratio) / fs) * cf0, &sd, &cd);
  }
  __device__ __forceinline__ void next() { kwy_rotate(c, s, cd, sd); }
  // Blackman: the second harmonic through the double-angle identity
  __device__ __forceinline__ double value(int type) const {
    return type == D4C_HANNING ? 0.5 * c + 0.5 : 0.42 + 0.5 * c + 0.08 * (2.0 * c * c - 1.0);
  }
};

// interp1Q with the grid step given as its reciprocal (two divisions per output bin saved;
// the position differs from (xi - x0) / shift by an ulp, the interpolant is continuous)
__device__ __forceinline__ double d4c_interp1q_inv(double x0, double inv_shift, const double *y, int x_length,
                                                   double xi) {
  double r = (xi - x0) * inv_shift;
  int base = (int)r;
  double frac = r - base;
  double y0 = y[base];
  double dy = (base >= x_length - 1) ? 0.0 : y[base + 1] - y0;
  return __builtin_fma(dy, frac, y0);
}

__device__ __forceinline__ double d4c_interp1q(double x0, double shift, const double *y, int x_length,
                                               double xi) {
  double r = (xi - x0) / shift;
  int base = (int)r;
  double frac = r - base;
  double y0 = y[base];
  double dy = (base >= x_length - 1) ? 0.0 : y[base + 1] - y0;
  return __builtin_fma(dy, frac, y0);
}

// WORLD DCCorrection in place on P[0..H]; S: scratch
template <int NT>
__device__ inline void d4c_dc_correction(double *P, double *S, double cf0, int fs, int N) {
  const int upper_limit = 2 + (int)(cf0 * N / fs);
  const int nrep = upper_limit - 1;
  const double shift = -(double)fs / N;
  for (int k = threadIdx.x; k < nrep; k += NT)
    S[k] = d4c_interp1q(cf0, shift, P, upper_limit + 1, (double)k * fs / N);
  __syncthreads();
  for (int k = threadIdx.x; k < nrep; k += NT) P[k] = P[k] + S[k];
  __syncthreads();
}

// The two interpolated reads of a smoothed bin, (S(f_k + w/2) - S(f_k - w/2)) / w on the cumulative spectrum S: the
// bins are equidistant, so position and weight of both reads are those of bin 0 shifted by k -- formed once per
// smoothing (uniform), not per bin: 4 LDS reads, 2 fused multiply-adds and a difference per bin instead of two
// position computations with their conversions (a tenth of k_d4c_body's instructions).  The weights differ from the
// per-bin ones by the rounding of k + const (~1e-13 of a weight).
struct d4c_taps { int c0, c1; double f0, f1; };
__device__ __forceinline__ d4c_taps d4c_smoothing_taps(double origin, double inv_dfi, double width) {
  const double r0 = (-width / 2.0 - origin) * inv_dfi, r1 = (-width / 2.0 + width - origin) * inv_dfi;
  d4c_taps t;
  t.c0 = __builtin_amdgcn_readfirstlane((int)r0);
  t.c1 = __builtin_amdgcn_readfirstlane((int)r1);
  t.f0 = kwy_uniform(r0 - (double)(int)r0);
  t.f1 = kwy_uniform(r1 - (double)(int)r1);
  return t;
}
__device__ __forceinline__ double d4c_smoothed_bin(const double *S, int L, const d4c_taps &t, int k) {
  const int b0 = k + t.c0, b1 = k + t.c1;
  const double l0 = S[b0], l1 = S[min(b0 + 1, L - 1)], h0 = S[b1], h1 = S[min(b1 + 1, L - 1)];
  const double low = __builtin_fma(l1 - l0, t.f0, l0), high = __builtin_fma(h1 - h0, t.f1, h0);
  return high - low;
}

// chunk bound of the smoothing's prefix sum: L <= H + 2 (H/2) + 1 = N + 1 values over NT = N / 16 threads
#define D4C_SMOOTH_CHUNK 17
// WORLD LinearSmoothing: in[0..H] -> out[0..H] (out may alias in); S: scratch of >= H+2b+1
// (Measured in round 3 and dropped: the same three passes -- mirrored fill, prefix sum, two interpolated reads per
// bin -- unrolled to their compile-time maxima with the prefix sum's chunk in registers, so that the LDS reads of a
// pass are issued together: k_d4c_body 0.343 ms against 0.333 ms per launch of two utterances.  The kernel is bound
// by instruction issue, the other resident workgroups already cover the LDS latency of these loops, and the
// predicated unrolled forms issue more instructions.)
// kmax: only out[0 .. kmax] are wanted (the mirrored fill and the prefix sum stop where those bins stop reading)
template <int NT>
__device__ inline void d4c_linear_smoothing(const double *in, double *out, double *S, double *tot,
                                            double width, int fs, int N, int kmax) {
  const int H = N / 2;
  int boundary = (int)(width * N / fs) + 1;
  if (boundary > H / 2) boundary = H / 2;  // LDS guard; outside WORLD's domain anyway
  const int L = min(H + boundary * 2 + 1, kmax + boundary * 2 + 4);
  kwy_block_cumsum_of<NT, D4C_SMOOTH_CHUNK>([&](int i) {
    double m;
    if (i < boundary) m = in[boundary - i];
    else if (i < H + boundary) m = in[i - boundary];
    else m = in[H - (i - (H + boundary))];
    return m * fs / N;
  }, S, L, tot);
  const double origin = -(boundary - 0.5) * fs / N;
  const double inv_dfi = (double)N / fs, inv_width = 1.0 / width;
  const d4c_taps taps = d4c_smoothing_taps(origin, inv_dfi, width);
  for (int k = threadIdx.x; k <= min(H, kmax); k += NT) out[k] = d4c_smoothed_bin(S, L, taps, k) * inv_width;
  __syncthreads();
}

// one utterance of a launch (kwy_internal.hpp: kwy_batch)
struct d4c_view {
  const double *x, *tpos, *f0;
  double *out;
  uint64_t *offs_lt;    // T + 1: stream positions of the LoveTrain windows; [T] = the draws of the whole pass
  uint64_t *offs_b;     // T + 1: scratch of the body scan
  uint64_t *offs3;      // 3 T: stream positions of the body's three windows per frame, relative to offs_lt[T]
  double *ap0;          // T: LoveTrain's voicing measure
  double *dvbuf;        // T x dv_stride: static group delay (the bins the band windows read), body -> bands
  int x_length, T;
};
typedef kwy_batch<d4c_view> d4c_batch;

// LoveTrain pass: per-frame draw counts and their exclusive prefix sums, one workgroup per utterance
__global__ __launch_bounds__(KWY_THREADS) void k_d4c_lt_scan(d4c_batch b, int fs) {
  __shared__ uint64_t tot[KWY_THREADS];
  const d4c_view v = b.u[blockIdx.x];
  const double *f0 = v.f0;
  kwy_block_count_scan<KWY_THREADS>([&](int64_t i) -> uint64_t {
    const double f = f0[i];
    if (f == 0.0) return 0;
    return (uint64_t)(kwy_matlab_round(1.5 * fs / (f > 40.0 ? f : 40.0)) * 2 + 1);
  }, v.T, v.offs_lt, tot);
}

// General body: per-frame draw counts (three windows; none for ungated frames), their prefix sums, and the draw
// offsets of the three windows of every frame (~0 = no work), one workgroup per utterance
__global__ __launch_bounds__(KWY_THREADS) void k_d4c_body_scan(d4c_batch b, int fs, double threshold) {
  __shared__ uint64_t tot[KWY_THREADS];
  const d4c_view v = b.u[blockIdx.x];
  const double *f0 = v.f0, *ap0 = v.ap0;
  uint64_t *offs = v.offs_b, *offs3 = v.offs3;
  const int64_t T = v.T;
  auto window = [&](int64_t i) -> uint64_t {      // draws of ONE window of frame i
    const double f = f0[i];
    if (f == 0.0 || ap0[i] <= threshold) return 0;
    return (uint64_t)(kwy_matlab_round(2.0 * fs / (f > D4C_FLOOR_F0 ? f : D4C_FLOOR_F0)) * 2 + 1);
  };
  kwy_block_count_scan<KWY_THREADS>([&](int64_t i) -> uint64_t { return 3 * window(i); }, T, offs, tot);
  __syncthreads();
  for (int64_t i = threadIdx.x; i < T; i += KWY_THREADS) {
    const uint64_t wl = window(i), o = offs[i];
#pragma unroll
    for (int w = 0; w < 3; ++w) offs3[3 * i + w] = wl ? o + (uint64_t)w * wl : ~0ull;
  }
}

// ------------------------------------------------------------------ LoveTrain
// One FFT buffer (in-place transform, pass factors in registers): 33 KB of LDS, four frames per CU.
// NT threads: 256, or 512 for the 8192-point transform of 96 kHz (one radix-8 butterfly per thread and pass).
template <int LOG2N, int NT>
__global__ __launch_bounds__(NT) void k_d4c_lovetrain(
    d4c_batch batch, int fs, kwy_randn_src rs,
    const uint4 *__restrict__ poly, const kwy_c *__restrict__ twH, const kwy_c *__restrict__ twN) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  constexpr int C = N / NT;
  constexpr int HEX = 16 * NT / N;
  constexpr int RK = (H + 1 + NT - 1) / NT;
  extern __shared__ double smem[];
  double *red = smem;                                 // 8
  uint32_t *e = (uint32_t *)(red + 8);                // KWY_EBASE_WORDS
  kwy_c *B = (kwy_c *)(e + KWY_EBASE_WORDS);          // H+1 complex (at least 8 KB: the RNG jump table)
  double *Bd = (double *)B;

  const int tid = threadIdx.x;
  const int utt = batch.find(blockIdx.x);
  const int64_t frame = (int)blockIdx.x - batch.start[utt];
  const double *__restrict__ x = batch.u[utt].x;
  const int x_length = batch.u[utt].x_length;
  const uint64_t *__restrict__ offsets = batch.u[utt].offs_lt;
  double *__restrict__ ap0 = batch.u[utt].ap0;
  const double f0v = batch.u[utt].f0[frame];
  if (f0v == 0.0) {
    if (tid == 0) ap0[frame] = 0.0;
    return;
  }
  const double cf0 = f0v > 40.0 ? f0v : 40.0;
  const int half = kwy_matlab_round(3.0 * fs / cf0 / 2.0);
  const int wl = 2 * half + 1;
  const int origin = kwy_matlab_round(batch.u[utt].tpos[frame] * fs + 0.001);
  kwy_c tw4[4];
  kwy_fft_thread_twiddles<LOG2N - 1, NT>(twH, tw4);
  const kwy_c twb = twN[tid];

  // the frame's wl noise draws (stream position offsets[frame]): sample i = tid + NT r takes draw i
  uint32_t raw[C];
  const uint64_t dpos = offsets[frame];
  if (dpos + (uint64_t)wl <= rs.n) {
#pragma unroll
    for (int r = 0; r < C; ++r) raw[r] = (tid + NT * r < wl) ? rs.tab[dpos + tid + NT * r] : 0u;
  } else {
    // beyond the table: thread t jumps to draw c t and makes c consecutive ones (c adapts to the window)
    kwy_rng_block_ebase(dpos, rs.pow2, e);
    const int c = (wl + NT - 1) / NT;
    kwy_rng_build_table<NT>(e, (uint4 *)B);
    __syncthreads();
    kwy_rng rng = kwy_rng_combine_table((const uint4 *)B, poly[(c - 1) * NT + tid]);
    __syncthreads();  // the table is consumed: the buffer takes the draws
    uint32_t *D = (uint32_t *)B;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if (j < c) {
        const uint32_t v = kwy_rng_randn_raw(rng);
        if (c * tid + j < wl) D[c * tid + j] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < C; ++r) raw[r] = (tid + NT * r < wl) ? D[tid + NT * r] : 0u;
    __syncthreads();
  }
  // sample i = tid + NT r: the window value goes to Bd[i], the sample waits in a register
  double vv[C];
  double s1 = 0.0, s2 = 0.0;
  d4c_window_walk walk(tid, NT, half, 3.0, fs, cf0);
#pragma unroll
  for (int r = 0; r < C; ++r) {
    const int i = tid + NT * r;
    double v = 0.0;
    if (r > 0) walk.next();
    if (i < wl) {
      double w = walk.value(D4C_BLACKMAN);
      int idx = min(x_length - 1, max(0, origin + i - half));
      v = __builtin_fma(kwy_randn_from_raw(raw[r]), D4C_SAFE, x[idx] * w);
      Bd[i] = w;
      s1 += v; s2 += w;
    }
    vv[r] = v;
  }
  const double t1 = kwy_block_sum<NT>(s1, red);
  const double t2 = kwy_block_sum<NT>(s2, red);
  const double coef = t1 / t2;
#pragma unroll
  for (int r = 0; r < C; ++r) {
    const int i = tid + NT * r;
    Bd[i] = (i < wl) ? vv[r] - Bd[i] * coef : 0.0;
  }
  __syncthreads();
  kwy_fft_inplace_w<LOG2N - 1, NT, false>(B, tw4);

  const int boundary0 = (int)ceil(100.0 * N / fs);
  const int boundary1 = (int)ceil(4000.0 * N / fs);
  const int boundary2 = (int)ceil(7900.0 * N / fs);
  double c1 = 0.0, c2 = 0.0;
#pragma unroll
  for (int r = 0; r < RK; ++r) {
    const int k = tid + NT * r;
    if (k > boundary0 && k <= boundary2 && k <= H) {
      const kwy_c v = kwy_rfft_bin2_w<LOG2N - 1>(B, k, kwy_tw_hex(twb, HEX * r));
      double pw = __builtin_fma(v.x, v.x, v.y * v.y);
      c2 += pw;
      if (k <= boundary1) c1 += pw;
    }
  }
  const double n1 = kwy_block_sum<NT>(c1, red);
  const double n2 = kwy_block_sum<NT>(c2, red);
  if (tid == 0) ap0[frame] = n1 / n2;
}

// ------------------------------------------------------------------ general body
struct d4c_params {
  int fs, K, fft_size, nbands, window_length;
  int dv_len;       // bins of the static group delay the band windows read: [0, dv_len) -- 1537 of 2049 at 48 kHz
  int dv_stride;    // row stride of dvbuf (dv_len rounded up to an even count)
  double threshold;
};

// One WORLD window of the frame: x around `pos`, times the window function, plus the
// safeguard noise, DC removed with the window as weight.  Element i = tid + NT*r
// stays in av[r] (0 beyond the window).  The window's wl noise draws are a contiguous piece
// of the serial stream starting at draw `dpos`: element i takes draw i, from the table; beyond the
// table thread t jumps to draw c*t and produces c = ceil(wl/NT) draws, which travel through
// Bd (N doubles of LDS, free on entry) to the threads that use them.  `normalise` scales to
// unit power (GetCentroid).
// wcos / wmode: the window's cosine c1(i) is the same for the frame's three windows (one f0, ratio 4 for the Blackman
// and the Hanning window alike, and exactly even around the centre), so the first window leaves it in LDS (wmode 1:
// slot min(i, wl - 1 - i) of wcos, <= N/4 + 1 doubles) and the other two read it (wmode 2) instead of evaluating the
// polynomial again -- ~45 instructions per element, a tenth of the kernel.  wmode 0: evaluate, no cache (the frames
// whose draws lie beyond the table build their jump table where the cache lives).
template <int N, int NT>
__device__ __forceinline__ void d4c_frame_window(const double *__restrict__ x, int x_length, const d4c_params &p, double cf0,
                                                 double pos, int type, uint64_t dpos, const kwy_randn_src &rs,
                                                 const uint4 *__restrict__ poly, uint32_t *e, uint4 *jtab,
                                                 double *Bd, bool normalise, double *red,
                                                 double (&av)[N / NT], double *wcos, int wmode) {
  constexpr int E = N / NT;
  const int tid = kwy_tid_opaque();
  // The window function is the same for the two centroid windows; left alone, the compiler
  // evaluates it once and carries 2 x E doubles per thread across the FFTs (through scratch
  // memory).  Re-evaluating it is far cheaper, so cf0 is made opaque here.
  asm volatile("" : "+s"(cf0));
  const int half = kwy_matlab_round(4.0 * p.fs / cf0 / 2.0);
  const int wl = 2 * half + 1;
  const int origin = kwy_matlab_round(pos * p.fs + 0.001);
  // samples and draws are independent loads: get both on their way first
  double xv[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = tid + NT * r;
    xv[r] = (i < wl) ? x[min(x_length - 1, max(0, origin + i - half))] : 0.0;
  }
  uint32_t raw[E];
  if (dpos + (uint64_t)wl <= rs.n) {
#pragma unroll
    for (int r = 0; r < E; ++r) raw[r] = (tid + NT * r < wl) ? rs.tab[dpos + tid + NT * r] : 0u;
  } else {
    kwy_rng_block_ebase(dpos, rs.pow2, e);
    kwy_rng_build_table<NT>(e, jtab);
    __syncthreads();
    const int c = (wl + NT - 1) / NT;  // <= E
    kwy_rng rng = kwy_rng_combine_table(jtab, poly[(c - 1) * NT + tid]);
    uint32_t *D = (uint32_t *)Bd;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      if (j < c) {
        const uint32_t v = kwy_rng_randn_raw(rng);
        if (c * tid + j < wl) D[c * tid + j] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) raw[r] = (tid + NT * r < wl) ? D[tid + NT * r] : 0u;
    __syncthreads();
  }
  // (the rotation walk of the LoveTrain window was measured here too: no gain -- 0.185 ms either way -- and the
  // walk's four doubles push the kernel into scratch; one polynomial per element stays)
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = tid + NT * r;
    double v = 0.0;
    if (i < wl) {
      const int slot = i <= half ? i : wl - 1 - i;
      double c1;
      if (wmode == 2) {
        c1 = wcos[slot];
      } else {
        c1 = kwy_cos_pi_range(KWY_PI * ((2.0 * (i - half) / 4.0) / p.fs) * cf0);
        if (wmode == 1) wcos[slot] = c1;
      }
      const double w = type == D4C_HANNING ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);
      v = __builtin_fma(kwy_randn_from_raw(raw[r]), D4C_SAFE, xv[r] * w);
      Bd[i] = w;                       // kept for the DC removal below (only this thread reads it)
      s1 += v; s2 += w;
    }
    av[r] = v;
  }
  double t1, t2;
  kwy_block_sum2<NT>(s1, s2, red, &t1, &t2);
  const double coef = t1 / t2;
  double pw = 0.0;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    if (tid + NT * r < wl) {
      double v = __builtin_fma(-Bd[tid + NT * r], coef, av[r]);
      av[r] = v;
      pw = __builtin_fma(v, v, pw);
    }
  }
  if (normalise) {
    // one division, then products (each element within an ulp of the CPU's quotient)
    const double isq = 1.0 / sqrt(kwy_block_sum<NT>(pw, red));
#pragma unroll
    for (int r = 0; r < E; ++r)
      if (tid + NT * r < wl) av[r] = av[r] * isq;
  }
}

// WORLD LinearSmoothing as above, but out[k] = in[k] - smoothed[k] (the last step of the static group delay)
template <int NT>
__device__ inline void d4c_subtract_smoothed(double *io, double *S, double *tot, double width, int fs, int N, int kmax) {
  const int H = N / 2;
  int boundary = (int)(width * N / fs) + 1;
  if (boundary > H / 2) boundary = H / 2;
  const int L = min(H + boundary * 2 + 1, kmax + boundary * 2 + 4);
  kwy_block_cumsum_of<NT, D4C_SMOOTH_CHUNK>([&](int i) {
    double m;
    if (i < boundary) m = io[boundary - i];
    else if (i < H + boundary) m = io[i - boundary];
    else m = io[H - (i - (H + boundary))];
    return m * fs / N;
  }, S, L, tot);
  const double origin = -(boundary - 0.5) * fs / N;
  const double inv_dfi = (double)N / fs, inv_width = 1.0 / width;
  const d4c_taps taps = d4c_smoothing_taps(origin, inv_dfi, width);
  for (int k = threadIdx.x; k <= min(H, kmax); k += NT) io[k] = io[k] - d4c_smoothed_bin(S, L, taps, k) * inv_width;
  __syncthreads();
}

// WORLD LinearSmoothing with the result left in registers: outv[r] = smoothed[tid + NT*r]
template <int NT, int RK>
__device__ inline void d4c_linear_smoothing_regs(const double *in, double (&outv)[RK], double *S, double *tot,
                                                 double width, int fs, int N) {
  const int H = N / 2;
  int boundary = (int)(width * N / fs) + 1;
  if (boundary > H / 2) boundary = H / 2;
  const int L = H + boundary * 2 + 1;
  kwy_block_cumsum_of<NT, D4C_SMOOTH_CHUNK>([&](int i) {
    double m;
    if (i < boundary) m = in[boundary - i];
    else if (i < H + boundary) m = in[i - boundary];
    else m = in[H - (i - (H + boundary))];
    return m * fs / N;
  }, S, L, tot);
  const double origin = -(boundary - 0.5) * fs / N;
  const double inv_dfi = (double)N / fs, inv_width = 1.0 / width;
  const d4c_taps taps = d4c_smoothing_taps(origin, inv_dfi, width);
#pragma unroll
  for (int r = 0; r < RK; ++r) {
    const int k = threadIdx.x + NT * r;
    outv[r] = 0.0;
    if (k <= H) outv[r] = fmax(d4c_smoothed_bin(S, L, taps, k) * inv_width, 0.0);   // a smoothed POWER spectrum (see kwy_cheaptrick.hip)
  }
  __syncthreads();
}

// a per-thread constant behind an optimisation barrier: what is derived from it (the nine bin
// twiddles, a few multiplications each) is recomputed at every use instead of living in -- at this
// register budget, being spilled from -- 36 registers for the whole kernel
__device__ __forceinline__ kwy_c d4c_opaque(kwy_c v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y));
  return v;
}

// General body.  One 256-thread workgroup per frame, and the frame's working set squeezed into
// 51 KB of LDS so that THREE frames share a CU (the kernel is bound by barrier latency and f64 issue,
// not by bandwidth: a third resident frame is worth more than wider workgroups):
//   A0 (H+2 doubles)   RNG jump table during the window phases; then the power spectrum P; then the
//                      static group delay Dv
//   B  (H+1 complex)   the one FFT buffer (in-place radix-8 transforms); also the smoothing scratch S
//                      and, in the band loop, the select histogram
// What would not fit stays in registers: the centroid sum across the power-spectrum phase, X1 across the
// second FFT of a centroid, the smoothed power spectrum, the Nuttall window, the FFT pass factors.
// The spectra themselves never go back to LDS: each thread pulls "its" bins k = tid + NT*r out of the
// packed half-length transform into registers.
// 96 kHz (N = 8192): 512 threads and 99 KB, one frame per CU.
template <int LOG2N>
struct d4c_nt { static constexpr int value = LOG2N >= 13 ? 512 : 256; };
template <int LOG2N>
static constexpr size_t d4c_body_lds() {
  constexpr int N = 1 << LOG2N, H = N / 2, NT = d4c_nt<LOG2N>::value;
  // tot | red | coarse | e | [jtab when it does not fit A0] | A0 | B (+2 doubles of smoothing overflow)
  return sizeof(double) * (NT + 16 + D4C_MAX_BANDS + 2) + sizeof(uint32_t) * KWY_EBASE_WORDS +
         (LOG2N >= 12 ? 0 : 8192) + sizeof(double) * ((H + 2) + (2 * H + 2) + 2) +
         // the select scratch lives in B; short transforms need room for it
         ((sizeof(uint32_t) * KWY_SELECT_WORDS(NT) > sizeof(double) * (2 * H + 4))
              ? sizeof(uint32_t) * KWY_SELECT_WORDS(NT) - sizeof(double) * (2 * H + 4) : 0);
}

template <int LOG2N>
__global__ __launch_bounds__(d4c_nt<LOG2N>::value, LOG2N >= 13 ? 1 : 3) void k_d4c_body(
    d4c_batch batch, d4c_params p, kwy_randn_src rs,
    const uint4 *__restrict__ poly, const kwy_c *__restrict__ twH, const kwy_c *__restrict__ twN,
    long long *__restrict__ dbg) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  constexpr int NT = d4c_nt<LOG2N>::value;
  constexpr int E = N / NT;                  // window elements per thread
  constexpr int RK = (H + 1 + NT - 1) / NT;  // spectrum bins per thread
  constexpr int HEX = 16 * NT / N;           // 16th-root index step of the bin twiddles
  static_assert(HEX >= 1, "workgroup narrower than N/16");
#define D4C_STAMP(n) do { if (dbg && threadIdx.x == 0 && blockIdx.x == (unsigned)dbg[63]) dbg[n] = clock64(); } while (0)
  extern __shared__ double smem[];
  double *tot = smem;                        // NT
  double *red = tot + NT;                    // 16
  double *coarse = red + 16;                 // D4C_MAX_BANDS + 2
  uint32_t *e = (uint32_t *)(coarse + D4C_MAX_BANDS + 2);  // KWY_EBASE_WORDS
  double *jt_own = (double *)(e + KWY_EBASE_WORDS);        // 1024 doubles unless the table fits A0
  double *A0 = jt_own + (LOG2N >= 12 ? 0 : 1024);          // H+2
  kwy_c *B = (kwy_c *)(A0 + (H + 2));        // H+1 complex (+2 doubles)
  double *Bd = (double *)B;
  double *P = A0, *Dv = A0;
  double *S = Bd;                            // <= 2H+3 doubles
  uint4 *jtab = (uint4 *)((LOG2N >= 12) ? A0 : jt_own);

  const int tid = threadIdx.x;
  const int utt = batch.find(blockIdx.x);
  const int64_t frame = (int)blockIdx.x - batch.start[utt];
  const double *__restrict__ x = batch.u[utt].x;
  const int x_length = batch.u[utt].x_length;
  const uint64_t *__restrict__ offs3 = batch.u[utt].offs3;
  const uint64_t *__restrict__ draws_before = batch.u[utt].offs_lt + batch.u[utt].T;
  double *__restrict__ dvbuf = batch.u[utt].dvbuf;
  const double f0v = kwy_uniform(batch.u[utt].f0[frame]);
  double *o = batch.u[utt].out + frame * p.K;
  if (f0v == 0.0 || kwy_uniform(batch.u[utt].ap0[frame]) <= p.threshold) {
    for (int k = tid; k < p.K; k += NT) o[k] = 1.0 - D4C_SAFE;
    return;
  }
  const double cf0 = kwy_uniform(f0v > D4C_FLOOR_F0 ? f0v : D4C_FLOOR_F0);
  const double pos = kwy_uniform(batch.u[utt].tpos[frame]);

  D4C_STAMP(0);
  kwy_c tw4[4];   // this thread's factor of every radix-8 pass
  kwy_fft_thread_twiddles<LOG2N - 1, NT>(twH, tw4);
  // exp(-2 pi i k / N) of "my" spectrum bins k = tid + NT*r is this times a 16th root of unity
  const kwy_c twb = twN[tid];
  // stream positions of the frame's three windows: the body's noise continues where the LoveTrain pass stopped
  const uint64_t dbase = *draws_before;
  const uint64_t dpos0 = dbase + offs3[3 * frame], dpos1 = dbase + offs3[3 * frame + 1], dpos2 = dbase + offs3[3 * frame + 2];
  // (the window cosine cache lives in A0, where a frame beyond the noise table builds its jump table: no cache then)
  const bool cache = dpos2 + (uint64_t)(2 * kwy_matlab_round(4.0 * p.fs / cf0 / 2.0) + 1) <= rs.n;

  D4C_STAMP(1);
  // ---- static centroid: two temporal centroids at pos -+ 0.25/f0, each Re(X2 conj X1) of the
  //      normalised window a (X1) and the window times its sample index, b = a (i + 1) (X2).
  // Round 5: the two REAL sequences travel as ONE complex sequence c = a + j b.  Its N-point transform Z splits into
  // the H-point transforms of the even and the odd samples, Z[k] = E[k] + W^k O[k], Z[H + k] = E[k] - W^k O[k], and
  //     Re(X2[k] conj X1[k]) = Im(Z[k] Z[N - k]) / 2
  // (X1 = (Z[k] + conj Z[N-k]) / 2, X2 = (Z[k] - conj Z[N-k]) / 2j): a thread that holds E and O at k = p and H - p
  // forms bins p and H - p with two complex products -- instead of four even/odd extractions of packed real
  // transforms (two per bin and transform) followed by the product.  Same two H-point transforms per centroid.
  // Pairs (p, H - p), p = tid + NT r < H/2, live in registers; the self-paired bin H/2 is thread 0's, through two
  // LDS slots (a fifth register set for one bin would tip the kernel into scratch).
  constexpr int RP = (H / 2 + NT - 1) / NT;
  double cen_lo[RP], cen_hi[RP];
  double *mid = coarse;            // [0..1]: E[H/2], [2]: the centroid sum of bin H/2  (the band values' slots: unused here)
  {
    double av[E];
    for (int which = 0; which < 2; ++which) {
      const int tid = kwy_tid_opaque();
      double cpos = which == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0;
      d4c_frame_window<N, NT>(x, x_length, p, cf0, cpos, D4C_BLACKMAN, which == 0 ? dpos0 : dpos1, rs, poly, e, jtab, Bd, true, red, av,
                              A0, cache ? 1 + which : 0);
      __syncthreads();      // (the window code keeps per-thread values in Bd until here)
      // even samples: z[m] = a[2m] + j b[2m] -- sample i = tid + NT r has the parity of tid
      if (!(tid & 1)) {
#pragma unroll
        for (int r = 0; r < E; ++r) { const int i = tid + NT * r; Bd[i] = av[r]; Bd[i + 1] = av[r] * (i + 1.0); }
      }
      __syncthreads();
      D4C_STAMP(2 + which * 2);
      kwy_fft_inplace_w<LOG2N - 1, NT, false>(B, tw4);
      kwy_c Ek[RP], Eh[RP];
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const int pp = tid + NT * r;
        Ek[r] = Eh[r] = kwy_c{0.0, 0.0};
        if (pp < H / 2) { Ek[r] = B[pp]; Eh[r] = B[(H - pp) & (H - 1)]; }     // (p = 0: E[H] = E[0])
      }
      if (tid == 0) { mid[0] = B[H / 2].x; mid[1] = B[H / 2].y; }
      __syncthreads();
      if (tid & 1) {
#pragma unroll
        for (int r = 0; r < E; ++r) { const int i = tid + NT * r; Bd[i - 1] = av[r]; Bd[i] = av[r] * (i + 1.0); }
      }
      __syncthreads();
      kwy_fft_inplace_w<LOG2N - 1, NT, false>(B, tw4);
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const int pp = tid + NT * r;
        if (pp < H / 2) {
          const kwy_c w = kwy_tw_hex(d4c_opaque(twb), HEX * r);            // W^p = exp(-2 pi i p / N)
          const kwy_c Ok = B[pp], Oh = B[(H - pp) & (H - 1)];
          const kwy_c t = cmulf(w, Ok);                                     // W^p O[p]
          const kwy_c u = cmulf(kwy_c{-w.x, w.y}, Oh);                      // W^(H-p) O[H-p] = -conj(W^p) O[H-p]
          const kwy_c zp = cadd(Ek[r], t), zhp = csub(Ek[r], t);            // Z[p], Z[H + p]
          const kwy_c zq = cadd(Eh[r], u), znp = csub(Eh[r], u);            // Z[H - p], Z[N - p]
          // 2 Im(Z[k] Z[N-k]) = 4 Re(X2 conj X1): the scale of the "times two" bins the power spectrum uses
          const double lo = 2.0 * __builtin_fma(zp.x, znp.y, zp.y * znp.x);
          const double hi = 2.0 * __builtin_fma(zq.x, zhp.y, zq.y * zhp.x);
          cen_lo[r] = which == 0 ? lo : cen_lo[r] + lo;
          cen_hi[r] = which == 0 ? hi : cen_hi[r] + hi;
        }
      }
      if (tid == 0) {     // bin H/2: W^(H/2) = -i, Z[H/2] = E + t, Z[N - H/2] = Z[H + H/2] = E - t
        const kwy_c Em = {mid[0], mid[1]}, Om = B[H / 2];
        const kwy_c t = {Om.y, -Om.x};
        const kwy_c za = cadd(Em, t), zb = csub(Em, t);
        const double v = 2.0 * __builtin_fma(za.x, zb.y, za.y * zb.x);
        mid[2] = which == 0 ? v : mid[2] + v;
      }
      __syncthreads();
      D4C_STAMP(3 + which * 2);
    }

    D4C_STAMP(6);
    // ---- smoothed power spectrum (the centroid sum waits in registers)
    d4c_frame_window<N, NT>(x, x_length, p, cf0, pos, D4C_HANNING, dpos2, rs, poly, e, jtab, Bd, false, red, av, A0, cache ? 2 : 0);
#pragma unroll
    for (int r = 0; r < E; ++r) Bd[tid + NT * r] = av[r];
    __syncthreads();
  }
  D4C_STAMP(7);
  kwy_fft_inplace_w<LOG2N - 1, NT, false>(B, tw4);
  D4C_STAMP(8);
  double pv[RK];
  {
    // the power spectrum in pairs (k, H - k), k = tid + NT q <= H/2 (kwy_rfft_pair_power2_w), H/2 by thread 0
    constexpr int Q = (H / 2) / NT;
    static_assert(RK == 2 * Q + 1, "pairs per thread");
#pragma unroll
    for (int q = 0; q < Q; ++q)
      kwy_rfft_pair_power2_w<LOG2N - 1>(B, tid + NT * q, kwy_tw_hex(d4c_opaque(twb), HEX * q), &pv[2 * q], &pv[2 * q + 1]);
    pv[2 * Q] = 0.0;
    if (tid == 0) {
      double pm;
      kwy_rfft_pair_power2_w<LOG2N - 1>(B, H / 2, kwy_c{0.0, -1.0}, &pv[2 * Q], &pm);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int k = tid + NT * q;
      P[k] = pv[2 * q];
      P[H - k] = pv[2 * q + 1];
    }
    if (tid == 0) P[H / 2] = pv[2 * Q];
  }
  __syncthreads();
  d4c_dc_correction<NT>(P, S, cf0, p.fs, N);
  d4c_linear_smoothing_regs<NT, RK>(P, pv, S, tot, cf0, p.fs, N);   // P is dead from here on

  D4C_STAMP(9);
  // ---- static group delay: the centroid sum moves into A0
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const int pp = tid + NT * r;
    if (pp < H / 2) { Dv[pp] = cen_lo[r]; Dv[H - pp] = cen_hi[r]; }
  }
  if (tid == 0) Dv[H / 2] = mid[2];
  __syncthreads();
  d4c_dc_correction<NT>(Dv, S, cf0, p.fs, N);
#pragma unroll
  for (int r = 0; r < RK; ++r)
    if (tid + NT * r <= H) Dv[tid + NT * r] = Dv[tid + NT * r] / pv[r];
  __syncthreads();
  // (only bins [0, dv_len) of the result are ever read -- by the band windows; the second smoothing of those bins
  // reads its input up to two of its own half-widths further)
  const int kdv = p.dv_len - 1;
  d4c_linear_smoothing<NT>(Dv, Dv, S, tot, cf0 / 2.0, p.fs, N, kdv + 2 * ((int)(cf0 * N / p.fs) + 1) + 4);
  d4c_subtract_smoothed<NT>(Dv, S, tot, cf0, p.fs, N, kdv);

  D4C_STAMP(10);
  // ---- hand the static group delay to the band items (k_d4c_bands): one row of H+1 doubles per frame
  {
    double *dv = dvbuf + (size_t)frame * p.dv_stride;
    for (int k = tid; k < p.dv_len; k += NT) dv[k] = Dv[k];
  }
  D4C_STAMP(15);
#undef D4C_STAMP
}

// The band half of D4C, one workgroup per frame: per 3 kHz band a Nuttall-windowed slice of the static group delay
// -> FFT -> power spectrum -> sum of the (H - boundary) smallest bins against the sum of all (radix select, no sort)
// -> coarse aperiodicity; then the band values are interpolated to the K output bins.  33 KB of LDS and no window /
// RNG state: four workgroups per CU instead of the body's three.  (One work item per (frame, band) with an atomic
// hand-off to a finishing item was measured: 0.32 ms per launch against 0.1 ms -- the per-item start-up loads and
// the device-scope atomics cost more than the finer balance gains.)
template <int LOG2N>
static constexpr size_t d4c_bands_lds() {
  constexpr int N = 1 << LOG2N, H = N / 2, NT = d4c_nt<LOG2N>::value;
  return sizeof(double) * (16 + D4C_MAX_BANDS + 2 + 2 * NT) + sizeof(double) * ((2 * H + 2) + 2) +
         ((sizeof(uint32_t) * KWY_SELECT_WORDS(NT) > sizeof(double) * (2 * H + 4))
              ? sizeof(uint32_t) * KWY_SELECT_WORDS(NT) - sizeof(double) * (2 * H + 4) : 0);
}

// SPARSE: the window fits the first H/8 (+1) packed points (compile-time, so that the two input paths do not share
// one register budget)
template <int LOG2N, bool SPARSE>
__global__ __launch_bounds__(d4c_nt<LOG2N>::value, LOG2N >= 13 ? 2 : 4) void k_d4c_bands(
    d4c_batch batch, d4c_params p, const kwy_c *__restrict__ twH,
    const kwy_c *__restrict__ twN, const double *__restrict__ nuttall, long long *__restrict__ dbg) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  constexpr int NT = d4c_nt<LOG2N>::value;
  constexpr int E = N / NT;
  constexpr int RK = (H + 1 + NT - 1) / NT;
  constexpr int HEX = 16 * NT / N;
#define D4C_STAMP(n) do { if (dbg && threadIdx.x == 0 && blockIdx.x == (unsigned)dbg[63]) dbg[n] = clock64(); } while (0)
  extern __shared__ double smem[];
  double *red = smem;                        // 16
  double *coarse = red + 16;                 // D4C_MAX_BANDS + 2
  double *wns = coarse + D4C_MAX_BANDS + 2;  // 2 NT: the thread's two window values (sparse path), parked between bands
  kwy_c *B = (kwy_c *)(wns + 2 * NT);        // H+1 complex (+2 doubles)
  double *Bd = (double *)B;
  uint32_t *hist = (uint32_t *)B;

  const int tid = threadIdx.x;
  const int utt = batch.find(blockIdx.x);
  const int64_t frame = (int)blockIdx.x - batch.start[utt];
  const double *__restrict__ dvbuf = batch.u[utt].dvbuf;
  double *__restrict__ out = batch.u[utt].out;
  const double f0v = kwy_uniform(batch.u[utt].f0[frame]);
  if (f0v == 0.0 || kwy_uniform(batch.u[utt].ap0[frame]) <= p.threshold) return;   // the body wrote the frame's row
  const double cf0 = kwy_uniform(f0v > D4C_FLOOR_F0 ? f0v : D4C_FLOOR_F0);
  const double *Dv = dvbuf + (size_t)frame * p.dv_stride;
  D4C_STAMP(16);

  // (uniform values formed in vector registers: moved to scalar ones, the kernel sits at its register cap)
  const int boundary = __builtin_amdgcn_readfirstlane(kwy_matlab_round(N * 8.0 / p.window_length));
  const int half_window_length = p.window_length / 2;
  constexpr bool sparse = SPARSE;
  // the Nuttall window stays in registers: elements 2 tid, 2 tid + 1 (and the last one) for the copy-only first
  // pass, elements tid + NT r otherwise
  double ns0 = 0.0, ns1 = 0.0, ns2 = 0.0, nutr[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (sparse) {
    if (2 * tid < p.window_length) ns0 = nuttall[2 * tid];
    if (2 * tid + 1 < p.window_length) ns1 = nuttall[2 * tid + 1];
    wns[2 * tid] = ns0; wns[2 * tid + 1] = ns1;      // (read back by the same thread: no barrier)
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (tid + NT * r < p.window_length) nutr[r] = nuttall[tid + NT * r];
  }
  // the group-delay slice of band b + 1 is fetched while band b is transformed and selected (its latency would
  // otherwise sit at the head of every band): in the sparse case a thread needs two values per band
  const bool mine0 = sparse && tid < H / 8 && 2 * tid < p.window_length;
  const bool mine1 = sparse && tid < H / 8 && 2 * tid + 1 < p.window_length;
  const double *Dfirst = Dv + ((int)(D4C_FREQ_INTERVAL * N / p.fs) - half_window_length);
  double nx0 = (mine0 && p.nbands > 0) ? Dfirst[2 * tid] : 0.0;
  double nx1 = (mine1 && p.nbands > 0) ? Dfirst[2 * tid + 1] : 0.0;
  for (int b = 0; b < p.nbands; ++b) {
    const int tid = kwy_tid_opaque();
    // the thread's pass factors and bin twiddle are fetched again for every band (five 16-byte loads from L1 / L2,
    // through the opaque thread index so that they are not hoisted) instead of living in 20 registers across the
    // selection, where the kernel is at its cap of 128
    kwy_c tw4[4];
    kwy_fft_thread_twiddles<LOG2N - 1, NT>(twH + (tid - (int)threadIdx.x), tw4);
    const kwy_c twb = twN[tid];
    const double *Dc = Dv + ((int)(D4C_FREQ_INTERVAL * (b + 1) * N / p.fs) - half_window_length);
    const double c0 = nx0, c1 = nx1;
    if (b == 1) D4C_STAMP(17);
    {
      const double *Dn = Dv + ((int)(D4C_FREQ_INTERVAL * (b + 2) * N / p.fs) - half_window_length);
      const bool more = b + 1 < p.nbands;
      nx0 = (mine0 && more) ? Dn[2 * tid] : 0.0;
      nx1 = (mine1 && more) ? Dn[2 * tid + 1] : 0.0;
    }
    if constexpr (sparse) {
      // only the first H/8 (+1) packed points are non-zero: the first pass needs no input buffer
      kwy_c a0 = {c0 * wns[2 * tid], c1 * wns[2 * tid + 1]}, a1 = {0.0, 0.0};
      if (tid == 0 && 2 * (H / 8) < p.window_length) a1.x = Dc[2 * (H / 8)] * nuttall[2 * (H / 8)];
      kwy_fft_pass8_first_sparse_core<LOG2N - 1, NT, false>(B, kwy_tw_reg{tw4[0]}, a0, a1);
      kwy_fft_inplace_rest_w<LOG2N - 1, NT, false>(B, tw4);
      if (b == 1) D4C_STAMP(18);
    } else {
#pragma unroll
      for (int r = 0; r < E; ++r) {
        const int j = tid + NT * r;
        Bd[j] = (r < 4 && j < p.window_length) ? Dc[j] * nutr[r < 4 ? r : 0] : 0.0;
      }
      __syncthreads();
      kwy_fft_inplace_w<LOG2N - 1, NT, false>(B, tw4);
    }
    // CPU: power spectrum, sort ascending, cumulative sum, ratio of the (H - boundary) smallest to all
    // bins k and H - k in pairs (k = tid + NT q <= H/2; the self-paired H/2 goes to thread 0): slot 2 q and 2 q + 1 of
    // every thread, the last slot thread 0's alone -- which keys a thread holds does not matter to the selection,
    // only that slot r of thread t is filled exactly when t + NT r <= H
    unsigned long long key[RK];
    constexpr int Q = (H / 2) / NT;
    static_assert(RK == 2 * Q + 1, "pairs per thread");
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double pk, pm;
      kwy_rfft_pair_power2_w<LOG2N - 1>(B, tid + NT * q, kwy_tw_hex(d4c_opaque(twb), HEX * q), &pk, &pm);
      key[2 * q] = (unsigned long long)__double_as_longlong(pk);
      key[2 * q + 1] = (unsigned long long)__double_as_longlong(pm);
    }
    key[2 * Q] = ~0ull;
    if (tid == 0) {
      double pk, pm;
      kwy_rfft_pair_power2_w<LOG2N - 1>(B, H / 2, kwy_c{0.0, -1.0}, &pk, &pm);
      key[2 * Q] = (unsigned long long)__double_as_longlong(pk);
    }
    __syncthreads();
    if (b == 1) D4C_STAMP(19);
    double nsmall, nall;
    kwy_block_smallest_sum<RK, NT>(key, H + 1, H - boundary, hist, red, &nsmall, &nall);
    if (b == 1) D4C_STAMP(20);
    if (dbg && tid == 0 && blockIdx.x == (unsigned)dbg[63])
      for (int q = 0; q < 12; ++q) dbg[24 + 12 * (b & 1) + q] = hist[2 * KWY_SELECT_BINS + q];
    if (tid == 0) {
      double cv = 10 * log10(nsmall / nall);
      cv = cv + (cf0 - 100) / 50.0;
      coarse[b + 1] = cv < 0.0 ? cv : 0.0;
    }
    __syncthreads();
  }
  D4C_STAMP(21);
  if (tid == 0) {
    coarse[0] = -60.0;
    coarse[p.nbands + 1] = -D4C_SAFE;
  }
  __syncthreads();
  // ---- interp1 of the (nbands+2)-point contour onto the K output bins
  double *o = out + frame * p.K;
  const int nn = p.nbands + 2;
  for (int k = tid; k < p.K; k += NT) {
    double xi = (double)k * p.fs / p.fft_size;
    // number of nodes <= xi: the nodes are the multiples of the band interval up to nbands, then fs / 2 (the quotient
    // of an exact multiple is exact, so the floor agrees with the comparisons it replaces)
    int seg = min((int)(xi / D4C_FREQ_INTERVAL) + 1, p.nbands + 1);
    if (seg < 1) seg = 1;
    if (seg > nn - 1) seg = nn - 1;
    double xa = (seg - 1 <= p.nbands) ? (seg - 1) * D4C_FREQ_INTERVAL : p.fs / 2.0;
    double xb = (seg <= p.nbands) ? seg * D4C_FREQ_INTERVAL : p.fs / 2.0;
    double sfrac = (xi - xa) / (xb - xa);
    double v = coarse[seg - 1] + sfrac * (coarse[seg] - coarse[seg - 1]);
    o[k] = exp10(v / 20.0);
  }
  D4C_STAMP(22);
#undef D4C_STAMP
}

// ------------------------------------------------------------------ host side
template <int LOG2N>
static int launch_lt(kwy_ctx *ctx, const d4c_batch &b, int fs) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  const kwy_c *twH, *twN;
  const uint4 *poly;
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N - 1, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N, &twN));
  constexpr int NT = d4c_nt<LOG2N>::value;
  KWY_TRY(kwy_get_poly_multi(ctx, N / NT, NT, &poly));
  size_t lds = sizeof(kwy_c) * (H + 1) + sizeof(double) * 8 + sizeof(uint32_t) * KWY_EBASE_WORDS;
  KWY_HIP(hipFuncSetAttribute((const void *)k_d4c_lovetrain<LOG2N, NT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_d4c_lovetrain", hipLaunchKernelGGL((k_d4c_lovetrain<LOG2N, NT>), dim3((unsigned)b.start[b.n]), dim3(NT), lds,
                     ctx->stream, b, fs, kwy_randn(ctx), poly, twH, twN));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

template <int LOG2N>
static int launch_body(kwy_ctx *ctx, const d4c_batch &b, const d4c_params &p, const double *nuttall) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  const kwy_c *twH, *twN;
  const uint4 *poly;
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N - 1, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N, &twN));
  constexpr int NT = d4c_nt<LOG2N>::value;
  KWY_TRY(kwy_get_poly_multi(ctx, N / NT, NT, &poly));
  const unsigned grid = (unsigned)b.start[b.n];
  size_t lds = d4c_body_lds<LOG2N>();
  KWY_HIP(hipFuncSetAttribute((const void *)k_d4c_body<LOG2N>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_d4c_body", hipLaunchKernelGGL(k_d4c_body<LOG2N>, dim3(grid), dim3(NT), lds, ctx->stream, b, p,
                     kwy_randn(ctx), poly, twH, twN, (long long *)ctx->dbg));
  const size_t lds_b = d4c_bands_lds<LOG2N>();
  if (p.window_length <= 2 * (H / 8) + 1) {
    KWY_HIP(hipFuncSetAttribute((const void *)k_d4c_bands<LOG2N, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    KWY_PROF(ctx, "k_d4c_bands", hipLaunchKernelGGL((k_d4c_bands<LOG2N, true>), dim3(grid), dim3(NT), lds_b, ctx->stream,
                       b, p, twH, twN, nuttall, (long long *)ctx->dbg));
  } else {
    KWY_HIP(hipFuncSetAttribute((const void *)k_d4c_bands<LOG2N, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    KWY_PROF(ctx, "k_d4c_bands", hipLaunchKernelGGL((k_d4c_bands<LOG2N, false>), dim3(grid), dim3(NT), lds_b, ctx->stream,
                       b, p, twH, twN, nuttall, (long long *)ctx->dbg));
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int get_nuttall(kwy_ctx *ctx, int window_length, const double **out) {
  std::string key = "nuttall:" + std::to_string(window_length);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    std::vector<double> h(window_length);
    for (int i = 0; i < window_length; ++i) {
      double tmp = i / (window_length - 1.0);
      h[i] = 0.355768 - 0.487396 * cos(2.0 * KWY_PI * tmp) + 0.144232 * cos(4.0 * KWY_PI * tmp) -
             0.012604 * cos(6.0 * KWY_PI * tmp);
    }
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * window_length));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(double) * window_length, hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  *out = it->second;
  return KWY_OK;
}

static int d4c_fft_size(int fs) {
  return (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / D4C_FLOOR_F0 + 1) / 0.69314718055994529));
}

static int d4c_nbands(int fs) {
  double lim = fs / 2.0 - D4C_FREQ_INTERVAL;
  if (lim > D4C_UPPER_LIMIT) lim = D4C_UPPER_LIMIT;
  const int nb = (int)(lim / D4C_FREQ_INTERVAL);
  return nb < 0 ? 0 : nb;
}

// The band windows (2 hw + 1 taps around bin (int)(3000 (b + 1) N / fs), hw = (int)(3000 N / fs)) read the bins
// [0, (int)(3000 nbands N / fs) + hw] of the static group delay: 1537 of 2049 at 48 kHz.
static int d4c_dv_len(int fs, int n4, int nbands) {
  if (nbands <= 0) return 0;
  const int hw = (int)(D4C_FREQ_INTERVAL * n4 / fs);
  const int len = (int)(D4C_FREQ_INTERVAL * nbands * n4 / fs) + hw + 1;
  return len < n4 / 2 + 1 ? len : n4 / 2 + 1;
}

static size_t d4c_scratch_bytes(int64_t T, int fs) {
  const int n4 = d4c_fft_size(fs);
  const size_t stride = (size_t)((d4c_dv_len(fs, n4, d4c_nbands(fs)) + 1) & ~1);
  return kwy_pad(sizeof(double) * (size_t)T * (stride > 0 ? stride : 2)) + 2 * kwy_pad(sizeof(uint64_t) * (T + 1)) +
         kwy_pad(sizeof(uint64_t) * 3 * T) + kwy_pad(sizeof(double) * T);
}

// device-pointer core for up to KWY_BATCH_MAX utterances; the per-utterance scratch comes from the arena
// (d4c_scratch_bytes each)
static int d4c_core(kwy_ctx *ctx, d4c_batch &b, int fs, double threshold, int fft_size) {
  const int n4 = d4c_fft_size(fs);
  const int nl = (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / 40.0 + 1) / 0.69314718055994529));
  const int l4 = kwy_ilog2(n4), ll = kwy_ilog2(nl);
  if (l4 < 10 || l4 > 13 || ll < 10 || ll > 13) {
    ctx->err = "d4c: sampling rate outside the supported range (8 kHz .. 96 kHz)";
    return KWY_EINVAL;
  }
  d4c_params p;
  p.fs = fs;
  p.fft_size = fft_size;
  p.K = fft_size / 2 + 1;
  double lim = fs / 2.0 - D4C_FREQ_INTERVAL;
  if (lim > D4C_UPPER_LIMIT) lim = D4C_UPPER_LIMIT;
  p.nbands = (int)(lim / D4C_FREQ_INTERVAL);
  if (p.nbands < 0) p.nbands = 0;
  if (p.nbands > 5) { ctx->err = "d4c: too many bands"; return KWY_EINVAL; }
  p.window_length = (int)(D4C_FREQ_INTERVAL * n4 / fs) * 2 + 1;
  p.dv_len = d4c_dv_len(fs, n4, p.nbands);
  p.dv_stride = (p.dv_len + 1) & ~1;
  p.threshold = threshold;

  b.start[0] = 0;
  for (int u = 0; u < b.n; ++u) {
    d4c_view &v = b.u[u];
    const size_t T = (size_t)v.T;
    v.offs_lt = kwy_arena<uint64_t>(ctx, T + 1);
    v.offs_b = kwy_arena<uint64_t>(ctx, T + 1);
    v.offs3 = kwy_arena<uint64_t>(ctx, 3 * T);
    v.ap0 = kwy_arena<double>(ctx, T);
    v.dvbuf = kwy_arena<double>(ctx, T * (size_t)(p.dv_stride > 0 ? p.dv_stride : 2));
    if (!v.dvbuf || !v.offs_lt || !v.offs_b || !v.offs3 || !v.ap0) { ctx->err = "d4c: scratch arena too small"; return KWY_ENOMEM; }
    b.start[u + 1] = b.start[u] + v.T;
  }
  const double *nuttall;
  KWY_TRY(get_nuttall(ctx, p.window_length, &nuttall));

  // LoveTrain pass
  hipLaunchKernelGGL(k_d4c_lt_scan, dim3(b.n), dim3(KWY_THREADS), 0, ctx->stream, b, fs);
  KWY_HIP(hipGetLastError());
  switch (ll) {
    case 10: KWY_TRY(launch_lt<10>(ctx, b, fs)); break;
    case 11: KWY_TRY(launch_lt<11>(ctx, b, fs)); break;
    case 12: KWY_TRY(launch_lt<12>(ctx, b, fs)); break;
    default: KWY_TRY(launch_lt<13>(ctx, b, fs)); break;
  }
  // general body; its noise continues where the LoveTrain pass stopped (offs_lt[T], read by the kernel)
  hipLaunchKernelGGL(k_d4c_body_scan, dim3(b.n), dim3(KWY_THREADS), 0, ctx->stream, b, fs, threshold);
  KWY_HIP(hipGetLastError());
  switch (l4) {
    case 10: return launch_body<10>(ctx, b, p, nuttall);
    case 11: return launch_body<11>(ctx, b, p, nuttall);
    case 12: return launch_body<12>(ctx, b, p, nuttall);
    default: return launch_body<13>(ctx, b, p, nuttall);
  }
}

static int d4c_one(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t, const double *f0, int64_t T,
                   double threshold, int fft_size, double *out) {
  d4c_batch b;
  b.n = 1;
  b.u[0] = d4c_view{x, t, f0, out, nullptr, nullptr, nullptr, nullptr, nullptr, (int)x_length, (int)T};
  return d4c_core(ctx, b, fs, threshold, fft_size);
}

static int d4c_check(kwy_ctx *ctx, const void *x, int64_t x_length, int fs, const void *t,
                     const void *f0, int64_t T, const void *out, int *fft_size) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !t || !f0 || !out || x_length <= 0 || T <= 0 || fs <= 0 || x_length > 0x7fffffff) {
    ctx->err = "d4c: bad argument";
    return KWY_EINVAL;
  }
  if (*fft_size <= 0) *fft_size = kwy_cheaptrick_fft_size(fs, 71.0);
  return KWY_OK;
}

extern "C" int kwy_d4c_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                           const double *f0, int64_t T, double threshold, int fft_size,
                           double *out) {
  KWY_TRY(d4c_check(ctx, x, x_length, fs, t, f0, T, out, &fft_size));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, d4c_scratch_bytes(T, fs)));
  return d4c_one(ctx, x, x_length, fs, t, f0, T, threshold, fft_size, out);
}

// pyworld.d4c for `count` utterances of one sampling rate, one grid over all their frames per kernel
// (kwy_cheaptrick_batch_dev's descriptors; out: T x (fft_size / 2 + 1) each)
extern "C" int kwy_d4c_batch_dev(kwy_ctx *ctx, const kwy_utterance *utts, int count, int fs, double threshold,
                                 int fft_size) {
  if (!ctx) return KWY_EINVAL;
  if (!utts || count < 1) { ctx->err = "d4c_batch: bad argument"; return KWY_EINVAL; }
  size_t scratch = 0;
  int64_t frames = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_utterance &q = utts[i];
    KWY_TRY(d4c_check(ctx, q.x, q.x_length, fs, q.temporal_positions, q.f0, q.f0_length, q.out, &fft_size));
    scratch += d4c_scratch_bytes(q.f0_length, fs);
    frames += q.f0_length;
  }
  if (frames > 0x7fffffff) { ctx->err = "d4c_batch: too many frames"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, scratch));
  for (int i0 = 0; i0 < count; i0 += KWY_BATCH_MAX) {
    d4c_batch b;
    b.n = count - i0 < KWY_BATCH_MAX ? count - i0 : KWY_BATCH_MAX;
    for (int u = 0; u < b.n; ++u) {
      const kwy_utterance &q = utts[i0 + u];
      b.u[u] = d4c_view{q.x, q.temporal_positions, q.f0, q.out, nullptr, nullptr, nullptr, nullptr, nullptr,
                        (int)q.x_length, (int)q.f0_length};
    }
    KWY_TRY(d4c_core(ctx, b, fs, threshold, fft_size));
  }
  return KWY_OK;
}

extern "C" int kwy_d4c(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                       const double *f0, int64_t T, double threshold, int fft_size, double *out) {
  KWY_TRY(d4c_check(ctx, x, x_length, fs, t, f0, T, out, &fft_size));
  KWY_HIP(hipSetDevice(ctx->device));
  for (int64_t i = 0; i < T; ++i)
    if (!(f0[i] >= 0.0 && f0[i] < fs * 0.2)) { ctx->err = "d4c: f0 must lie in [0, fs/5)"; return KWY_EINVAL; }
  const int K = fft_size / 2 + 1;
  size_t bx = kwy_pad(sizeof(double) * x_length), bt = kwy_pad(sizeof(double) * T);
  size_t bo = kwy_pad(sizeof(double) * T * K);
  KWY_TRY(kwy_arena_begin(ctx, d4c_scratch_bytes(T, fs) + bx + 2 * bt + bo));
  double *dx = kwy_arena<double>(ctx, x_length), *dt = kwy_arena<double>(ctx, T);
  double *df0 = kwy_arena<double>(ctx, T), *dout = kwy_arena<double>(ctx, (size_t)T * K);
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * x_length, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dt, t, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(df0, f0, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(d4c_one(ctx, dx, x_length, fs, dt, df0, T, threshold, fft_size, dout));
  KWY_HIP(hipMemcpyAsync(out, dout, sizeof(double) * T * K, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
