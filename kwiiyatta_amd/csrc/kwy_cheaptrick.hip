// kwy_cheaptrick.hip -- CheapTrick spectral envelope on gfx950.
//
// Replaces pyworld.cheaptrick (reference call site kwiiyatta/vocoder/world.py:45;
// algorithm: Morise 2015, as shipped with pyworld 0.2.8).  One 256-thread
// workgroup per analysis frame; the whole frame lives in LDS:
//
//   F0-adaptive Hanning window of x (coalesced HBM reads of the framed audio)
//   -> real FFT (half-length complex Stockham FFT in LDS) -> power spectrum
//   -> DC correction -> linear smoothing (mirrored cumulative sum + 2 lerps)
//   -> + |randn|*eps -> log -> FFT -> smoothing/recovery lifter -> inverse FFT
//   -> exp -> HBM (coalesced, K doubles per frame)
//
// Algorithmic HBM bytes per frame: hop*8 (audio) + 16 (f0, t) in, K*8 out.
// The serial xorshift "randn" stream of the CPU algorithm is a constant: a frame
// loads its draws from the device's table of it (jump-ahead beyond the table;
// see kwy_device.hpp, kwy_ctx.hip).
#include "kwy_internal.hpp"

#define CT_EPS 0.00000000000000022204460492503131
#define CT_DEFAULT_F0 500.0

// one utterance of a launch
struct ct_view {
  const double *x, *tpos, *f0;
  double *out;           // T x K spectral envelope -- or, in the mel-cepstrum form, T x (order + 1) coefficients
  uint64_t *offsets;     // T + 1: stream position of every frame's first draw (WORLD reseeds per call: 0 for frame 0)
  int x_length, T;
};
// the mel-cepstrum form (k_cheaptrick<.., true> + k_cep2mc): the liftered cepstra of all frames of the launch
struct ct_mcep {
  double *cep;           // (frames of the launch) x ncs, in workgroup order
  int ncut, ncs;         // cepstral coefficients the frequency transform reads; row length (ncut rounded up to 4)
};
typedef kwy_batch<ct_view> ct_batch;

// per-frame draw counts (window length + K) and their exclusive prefix sums; one workgroup per utterance
__global__ __launch_bounds__(KWY_THREADS) void k_ct_scan(ct_batch b, int fs, double f0_floor_eff, int K) {
  __shared__ uint64_t tot[KWY_THREADS];
  const ct_view v = b.u[blockIdx.x];
  const double *f0 = v.f0;
  kwy_block_count_scan<KWY_THREADS>([&](int64_t i) -> uint64_t {
    const double cf0 = f0[i] <= f0_floor_eff ? CT_DEFAULT_F0 : f0[i];
    return (uint64_t)(2 * kwy_matlab_round(1.5 * fs / cf0) + 1 + K);
  }, v.T, v.offsets, tot);
}

// y(x) on the regular grid x0 + shift*j (WORLD interp1Q); delta of the last node is 0
__device__ __forceinline__ double ct_interp1q(double x0, double shift, const double *y, int x_length,
                                              double xi) {
  double r = (xi - x0) / shift;
  int base = (int)r;
  double frac = r - base;
  double y0 = y[base];
  double dy = (base >= x_length - 1) ? 0.0 : y[base + 1] - y0;
  return __builtin_fma(dy, frac, y0);
}

// MCEP: stop at the liftered cepstrum.  The smoothed log-envelope is log P[n] = sum_k c[k] cos(2 pi k n / N) over the
// symmetric cepstrum c the lifter leaves in the buffer -- exactly the cepstrum pysptk.sp2mc would recover from the
// envelope with an inverse transform of its logarithm (kwiiyatta/vocoder/mcep.py:68-71).  A consumer that only wants
// the mel-cepstrum therefore needs neither this kernel's last transform, its exp and its K-bin row, nor sp2mc's
// logarithm and transform of that row: the first ncut coefficients (c[0] halved, minus log(out_div) for the
// division of world.py:50) go to a scratch row and k_cep2mc applies the frequency transform as a matrix product.
template <int LOG2N, bool MCEP>
__global__ __launch_bounds__(KWY_THREADS, LOG2N <= 11 ? 5 : 4) void k_cheaptrick(
    ct_batch batch, int fs, double q1, double f0_floor_eff, kwy_randn_src rs, const uint4 *__restrict__ poly,
    const kwy_c *__restrict__ twH, const kwy_c *__restrict__ twN, double out_div, ct_mcep mcep) {
  constexpr int N = 1 << LOG2N;
  constexpr int H = N / 2;
  constexpr int K = H + 1;
  constexpr int E = N / KWY_THREADS;                           // window samples per thread: i = tid + 256 j
  constexpr int RK = (K + KWY_THREADS - 1) / KWY_THREADS;      // spectrum bins per thread: k = tid + 256 r
  constexpr int C = (N + K + KWY_THREADS - 1) / KWY_THREADS;   // jump-ahead path: consecutive draws per thread

  // LDS: one FFT buffer (in-place radix-8 transforms; it also serves as the smoothing scratch and
  // the log-spectrum), the power spectrum, a small twiddle table: 29 KB at fft 2048
  constexpr int TWL = (H / 8 > 1) ? H / 8 : 1;
  extern __shared__ double smem[];
  kwy_c *bufA = (kwy_c *)smem;           // H+1 complex
  double *P = (double *)(bufA + (H + 1));  // K (+1 pad)
  kwy_c *twl = (kwy_c *)(P + K + 1);     // exp(-2 pi i k / H), k < H/8
  double *red = (double *)(twl + TWL);   // 8
  double *tot = red + 8;                 // KWY_THREADS
  uint32_t *e = (uint32_t *)(tot + KWY_THREADS);  // KWY_EBASE_WORDS

  const int tid = threadIdx.x;
  const int utt = batch.find(blockIdx.x);
  const int64_t frame = (int)blockIdx.x - batch.start[utt];
  const double *__restrict__ x = batch.u[utt].x;
  const int x_length = batch.u[utt].x_length;
  const uint64_t *__restrict__ offsets = batch.u[utt].offsets;
  double *__restrict__ out = batch.u[utt].out;
  const double f0v = batch.u[utt].f0[frame];
  const double cf0 = f0v <= f0_floor_eff ? CT_DEFAULT_F0 : f0v;
  const double pos = batch.u[utt].tpos[frame];
  const int half = kwy_matlab_round(1.5 * fs / cf0);
  const int wl = 2 * half + 1;
  const int origin = kwy_matlab_round(pos * fs + 0.001);

  for (int i = tid; i < TWL; i += KWY_THREADS) twl[i] = twH[i];
  const kwy_c twb = twN[tid];
  // ---- noise: the frame consumes wl + K draws of the stream, from position offsets[frame]: draw i goes to window
  //      sample i, draw wl + k to bin k.  They come from the table (coalesced dwords) ...
  const uint64_t dpos = offsets[frame];
  uint32_t rw[E], rb[RK];
  if (dpos + (uint64_t)(wl + K) <= rs.n) {
    const uint32_t *tp = rs.tab + dpos;
#pragma unroll
    for (int j = 0; j < E; ++j) rw[j] = (tid + KWY_THREADS * j < wl) ? tp[tid + KWY_THREADS * j] : 0u;
#pragma unroll
    for (int r = 0; r < RK; ++r) rb[r] = (tid + KWY_THREADS * r <= H) ? tp[wl + tid + KWY_THREADS * r] : 0u;
  } else {
    // ... or, beyond it, from the generator itself: thread t jumps to draw c t and makes c consecutive ones, which
    // travel through LDS (the still unused FFT buffer) to the threads that use them
    kwy_rng_block_ebase(dpos, rs.pow2, e);
    const int c = (wl + K + KWY_THREADS - 1) / KWY_THREADS;    // <= C
    kwy_rng rng;
    if constexpr (sizeof(double) * (K + 1) >= 8192) {  // table-driven jump, table in the not yet used P array
      kwy_rng_build_table<KWY_THREADS>(e, (uint4 *)P);
      __syncthreads();
      rng = kwy_rng_combine_table((const uint4 *)P, poly[(c - 1) * KWY_THREADS + tid]);
    } else {
      rng = kwy_rng_combine(e, poly[(c - 1) * KWY_THREADS + tid]);
    }
    uint32_t *D = (uint32_t *)bufA;                            // wl + K <= N + K words: fits (H+1) complex
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if (j < c) {
        const uint32_t raw = kwy_rng_randn_raw(rng);
        if (c * tid + j < wl + K) D[c * tid + j] = raw;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; ++j) rw[j] = (tid + KWY_THREADS * j < wl) ? D[tid + KWY_THREADS * j] : 0u;
#pragma unroll
    for (int r = 0; r < RK; ++r) rb[r] = (tid + KWY_THREADS * r <= H) ? D[wl + tid + KWY_THREADS * r] : 0u;
    __syncthreads();
  }

  // ---- F0-adaptive window (thread owns samples i = tid + 256 j < wl: coalesced reads of the framed audio)
  // The Hanning window's argument advances by a constant from one j to the next: cos and sin of the first element,
  // then rotations by the step angle (kwy_device.hpp) instead of a trigonometric polynomial per element.
  double wv[E];
  double sumsq = 0.0;
  double wc, ws, wcd, wsd;
  kwy_sincos_pi_range(KWY_PI * ((tid - half) / 1.5 / fs) * cf0, &ws, &wc);   // |argument| <= pi inside the window
  kwy_sincos_medium(KWY_PI * (KWY_THREADS / 1.5 / fs) * cf0, &wsd, &wcd);
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid + KWY_THREADS * j;
    double w = 0.0;
    if (j > 0) kwy_rotate(wc, ws, wcd, wsd);
    if (i < wl) {
      w = 0.5 * wc + 0.5;
      sumsq += w * w;
    }
    wv[j] = w;
  }
  const double average = sqrt(kwy_block_sum(sumsq, red));
  double *A = (double *)bufA;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid + KWY_THREADS * j;
    if (i < wl) {
      double wn = wv[j] / average;
      wv[j] = wn;
      int idx = min(x_length - 1, max(0, origin + i - half));
      double v = x[idx] * wn;
      v = v + kwy_randn_from_raw(rw[j]) * 0.000000000000001;
      A[i] = v;
      s1 += v;
      s2 += wn;
    }
  }
  const double t1 = kwy_block_sum(s1, red);
  const double t2 = kwy_block_sum(s2, red);
  const double coef = t1 / t2;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid + KWY_THREADS * j;
    if (i < wl) A[i] -= wv[j] * coef;
  }
  for (int i = wl + tid; i < N; i += KWY_THREADS) A[i] = 0.0;

  // ---- power spectrum
  __syncthreads();
  kwy_rfft_inplace<LOG2N - 1, KWY_THREADS>(bufA, twl, twb, twN);
  double *S = (double *)bufA;   // scratch after P is formed (N+2 doubles)
  for (int k = tid; k <= H; k += KWY_THREADS) {
    kwy_c v = bufA[k];
    P[k] = v.x * v.x + v.y * v.y;
  }
  __syncthreads();

  // ---- DC correction
  {
    const int upper_limit = 2 + (int)(cf0 * N / fs);
    const int nrep = upper_limit - 1;
    const double shift = -(double)fs / N;
    for (int k = tid; k < nrep; k += KWY_THREADS)
      S[k] = ct_interp1q(cf0, shift, P, upper_limit + 1, (double)k * fs / N);
    __syncthreads();
    for (int k = tid; k < nrep; k += KWY_THREADS) P[k] = P[k] + S[k];
    __syncthreads();
  }

  // ---- linear smoothing, width 2 f0 / 3
  {
    const double width = cf0 * 2.0 / 3.0;
    int boundary = (int)(width * N / fs) + 1;
    if (boundary > H / 2) boundary = H / 2;  // guards LDS only; f0 > 3fs/8 is outside WORLD's domain
    const int L = H + boundary * 2 + 1;
    // (L <= N + 1 values over 256 threads: chunks of <= N / 256 + 1)
    kwy_block_cumsum_of<KWY_THREADS, N / KWY_THREADS + 1>([&](int i) {
      double m;
      if (i < boundary) m = P[boundary - i];
      else if (i < H + boundary) m = P[i - boundary];
      else m = P[H - (i - (H + boundary))];
      return m * fs / N;
    }, S, L, tot);
    const double origin_of_mirroring_axis = -(boundary - 0.5) * fs / N;
    const double dfi = (double)fs / N;
    // The bins are equidistant: position and weight of a bin's two interpolated reads are those of bin 0 shifted by k,
    // formed once (uniform) instead of per bin with a division each (kwy_d4c.hip: d4c_smoothing_taps).
    const double rl = (-width / 2.0 - origin_of_mirroring_axis) / dfi, rh = (-width / 2.0 + width - origin_of_mirroring_axis) / dfi;
    const int cl = __builtin_amdgcn_readfirstlane((int)rl), ch = __builtin_amdgcn_readfirstlane((int)rh);
    const double fl = kwy_uniform(rl - (double)(int)rl), fh = kwy_uniform(rh - (double)(int)rh);
    for (int k = tid; k <= H; k += KWY_THREADS) {
      const int b0 = k + cl, b1 = k + ch;
      const double l0 = S[b0], l1 = S[min(b0 + 1, L - 1)], h0 = S[b1], h1 = S[min(b1 + 1, L - 1)];
      const double low = __builtin_fma(l1 - l0, fl, l0), high = __builtin_fma(h1 - h0, fh, h0);
      // the serial CPU cumulative sum is monotone, so its differences are >= 0; the block-parallel one can
      // come out an ulp of the running total below zero in bins that carry no energy at all
      P[k] = fmax((high - low) / width, 0.0);
    }
    __syncthreads();
  }

  // ---- infinitesimal noise: draw wl + k belongs to bin k
#pragma unroll
  for (int r = 0; r < RK; ++r) {
    const int k = tid + KWY_THREADS * r;
    if (k <= H) P[k] = P[k] + fabs(kwy_randn_from_raw(rb[r])) * CT_EPS;
  }
  __syncthreads();

  // ---- log spectrum, mirrored, into the other buffer
  double *Lg = S;  // N doubles in buffer X
  for (int k = tid; k <= H; k += KWY_THREADS) {
    double v = kwy_log(P[k]);
    Lg[k] = v;
    if (k >= 1 && k < H) Lg[N - k] = v;
  }
  __syncthreads();
  kwy_rfft_inplace<LOG2N - 1, KWY_THREADS>(bufA, twl, twb, twN);
  kwy_c *Cx = bufA;

  // ---- smoothing + recovery lifter.  The lifters' argument pi cf0 k / fs advances by a constant from one of the
  //      thread's bins k = tid + 256 r to the next: one sincos for r = 0 and one for the step, then rotations
  //      (kwy_rotate) -- the general-range sin() per bin was 5 % of the kernel's instructions.
  {
    double ls, lc, lsd, lcd;
    kwy_sincos_medium(KWY_PI * cf0 * ((double)tid / fs), &ls, &lc);
    kwy_sincos_medium(KWY_PI * cf0 * ((double)KWY_THREADS / fs), &lsd, &lcd);
#pragma unroll
    for (int r = 0; r < RK; ++r) {
      const int k = tid + KWY_THREADS * r;
      if (r > 0) kwy_rotate(lc, ls, lcd, lsd);
      if (k <= H) {
        double sl, cl;
        if (k == 0) {
          sl = 1.0;
          cl = (1.0 - 2.0 * q1) + 2.0 * q1;
        } else {
          // one sine serves both factors: cos(2 th) = 1 - 2 sin^2(th) (differs from cos() by ~2 ulp of a factor near 1)
          const double th = KWY_PI * cf0 * ((double)k / fs);
          sl = ls / th;
          cl = (1.0 - 2.0 * q1) + 2.0 * q1 * (1.0 - 2.0 * ls * ls);
        }
        const double cv = Cx[k].x * sl * cl / N;
        if constexpr (MCEP) {
          if (k < mcep.ncs) {
            double *crow = mcep.cep + (size_t)blockIdx.x * mcep.ncs;
            crow[k] = k >= mcep.ncut ? 0.0 : (k == 0 ? (cv - log(out_div)) / 2.0 : cv);
          }
        } else {
          Cx[k] = {cv, 0.0};
        }
      }
    }
  }
  if constexpr (MCEP) {
    for (int k = H + 1 + tid; k < mcep.ncs; k += KWY_THREADS) mcep.cep[(size_t)blockIdx.x * mcep.ncs + k] = 0.0;   // row padding
    return;
  }
  kwy_irfft_inplace<LOG2N - 1, KWY_THREADS>(bufA, twl, twb, twN);
  const double *wr = (const double *)bufA;
  double *o = out + frame * K;
  if (out_div == 1.0) {
    for (int k = tid; k <= H; k += KWY_THREADS) o[k] = exp(wr[k]);
  } else {
    for (int k = tid; k <= H; k += KWY_THREADS) o[k] = exp(wr[k]) / out_div;
  }
}

// mc[t][j] = sum_n cep[t][n] F[n][j]: pysptk's frequency transform (freqt) of the cepstra as an f64 matrix product, one
// wavefront per 16 frames (v_mfma_f64_16x16x4_f64; A = 16 frames x 4 cepstral indices, B = F, 16 coefficients per
// block).  F: [ncut][64] (kwy_mcep.hip), L2-resident.  Row t of the launch belongs to the utterance batch.find(t).
typedef double ct_v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(KWY_THREADS) void k_cep2mc(ct_batch batch, ct_mcep mcep, int order,
                                                       const double *__restrict__ F) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ar = lane & 15, ak = lane >> 4;
  const int rows = batch.start[batch.n];
  const int t0 = ((int)blockIdx.x * KWY_WAVES + wv) * 16;
  if (t0 >= rows) return;
  const int nblk = (order + 16) / 16;                  // 16-coefficient blocks: <= 4
  const double *__restrict__ crow = mcep.cep + (size_t)min(t0 + ar, rows - 1) * mcep.ncs;
  ct_v4f64 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = ct_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int ks = 0; ks < mcep.ncs / 4; ++ks) {
    const int n = 4 * ks + ak;
    const double a = crow[n];
    const double *__restrict__ frow = F + (size_t)min(n, mcep.ncut - 1) * 64 + ar;    // (rows beyond ncut: a == 0)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (b < nblk) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, frow[16 * b], acc[b], 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = t0 + ak + 4 * r;
    if (t >= rows) continue;
    int u = 0;
    while (u + 1 < batch.n && t >= batch.start[u + 1]) ++u;
    double *o = batch.u[u].out + (size_t)(t - batch.start[u]) * (order + 1);
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (b < nblk && 16 * b + ar <= order) o[16 * b + ar] = acc[b][r];
  }
}

template <int LOG2N, bool MCEP>
static int launch_ct(kwy_ctx *ctx, const ct_batch &b, int fs, double q1, double floor_eff, double out_div, ct_mcep mcep) {
  constexpr int N = 1 << LOG2N, H = N / 2, K = H + 1;
  constexpr int C = (N + K + KWY_THREADS - 1) / KWY_THREADS;
  const kwy_c *twH, *twN;
  const uint4 *poly;
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N - 1, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N, &twN));
  KWY_TRY(kwy_get_poly_multi(ctx, C, KWY_THREADS, &poly));
  size_t lds = sizeof(kwy_c) * ((H + 1) + (H / 8 > 1 ? H / 8 : 1)) + sizeof(double) * (K + 1 + 8 + KWY_THREADS) +
               sizeof(uint32_t) * KWY_EBASE_WORDS;
  KWY_HIP(hipFuncSetAttribute((const void *)k_cheaptrick<LOG2N, MCEP>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_cheaptrick", hipLaunchKernelGGL((k_cheaptrick<LOG2N, MCEP>), dim3((unsigned)b.start[b.n]), dim3(KWY_THREADS), lds,
                     ctx->stream, b, fs, q1, floor_eff, kwy_randn(ctx), poly, twH, twN, out_div, mcep));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static size_t ct_scratch_bytes(int64_t T) {
  return kwy_pad(sizeof(uint64_t) * (T + 1));
}

// device-pointer core for up to KWY_BATCH_MAX utterances (b.u[].offsets are filled in here from the arena, which must
// already have room: ct_scratch_bytes per utterance).  mc_order >= 0: the mel-cepstrum form (b.u[].out: T x
// (mc_order + 1)); the arena then also holds the launch's cepstra (ct_mcep_bytes).
static size_t ct_mcep_bytes(int64_t frames, int ncs) { return kwy_pad(sizeof(double) * (size_t)frames * ncs); }

static int cheaptrick_core(kwy_ctx *ctx, ct_batch &b, int fs, double q1, int fft_size, double out_div, int mc_order = -1,
                           double mc_alpha = 0.0) {
  const int log2n = kwy_ilog2(fft_size);
  if ((1 << log2n) != fft_size || log2n < 9 || log2n > 12) {
    ctx->err = "cheaptrick: fft_size must be a power of two in [512, 4096]";
    return KWY_EINVAL;
  }
  const int K = fft_size / 2 + 1;
  const double floor_eff = 3.0 * fs / (fft_size - 3.0);
  b.start[0] = 0;
  for (int u = 0; u < b.n; ++u) {
    b.u[u].offsets = kwy_arena<uint64_t>(ctx, (size_t)b.u[u].T + 1);
    if (!b.u[u].offsets) { ctx->err = "cheaptrick: scratch arena too small"; return KWY_ENOMEM; }
    b.start[u + 1] = b.start[u] + b.u[u].T;
  }
  hipLaunchKernelGGL(k_ct_scan, dim3(b.n), dim3(KWY_THREADS), 0, ctx->stream, b, fs, floor_eff, K);
  KWY_HIP(hipGetLastError());
  if (mc_order < 0) {
    const ct_mcep none = {nullptr, 0, 0};
    switch (log2n) {
      case 9: return launch_ct<9, false>(ctx, b, fs, q1, floor_eff, out_div, none);
      case 10: return launch_ct<10, false>(ctx, b, fs, q1, floor_eff, out_div, none);
      case 11: return launch_ct<11, false>(ctx, b, fs, q1, floor_eff, out_div, none);
      default: return launch_ct<12, false>(ctx, b, fs, q1, floor_eff, out_div, none);
    }
  }
  const double *F;
  ct_mcep mcep;
  KWY_TRY(kwy_get_sp2mc_matrix(ctx, fft_size, mc_order, mc_alpha, &F, &mcep.ncut));
  if (mcep.ncut > K) mcep.ncut = K;
  mcep.ncs = (mcep.ncut + 3) & ~3;
  mcep.cep = kwy_arena<double>(ctx, (size_t)b.start[b.n] * mcep.ncs);
  if (!mcep.cep) { ctx->err = "cheaptrick: scratch arena too small"; return KWY_ENOMEM; }
  switch (log2n) {
    case 9: KWY_TRY((launch_ct<9, true>(ctx, b, fs, q1, floor_eff, out_div, mcep))); break;
    case 10: KWY_TRY((launch_ct<10, true>(ctx, b, fs, q1, floor_eff, out_div, mcep))); break;
    case 11: KWY_TRY((launch_ct<11, true>(ctx, b, fs, q1, floor_eff, out_div, mcep))); break;
    default: KWY_TRY((launch_ct<12, true>(ctx, b, fs, q1, floor_eff, out_div, mcep))); break;
  }
  const unsigned tiles = (unsigned)((b.start[b.n] + 15) / 16);
  KWY_PROF(ctx, "k_cep2mc", hipLaunchKernelGGL(k_cep2mc, dim3((tiles + KWY_WAVES - 1) / KWY_WAVES), dim3(KWY_THREADS), 0,
                                               ctx->stream, b, mcep, mc_order, F));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int cheaptrick_one(kwy_ctx *ctx, const double *x, int64_t x_length, int fs, const double *t,
                          const double *f0, int64_t T, double q1, int fft_size, double out_div, double *out) {
  ct_batch b;
  b.n = 1;
  b.u[0] = ct_view{x, t, f0, out, nullptr, (int)x_length, (int)T};
  return cheaptrick_core(ctx, b, fs, q1, fft_size, out_div);
}

static int ct_check(kwy_ctx *ctx, const void *x, int64_t x_length, int fs, const void *t,
                    const void *f0, int64_t T, const void *out, int *fft_size, double f0_floor) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !t || !f0 || !out || x_length <= 0 || T <= 0 || fs <= 0 || x_length > 0x7fffffff) {
    ctx->err = "cheaptrick: bad argument";
    return KWY_EINVAL;
  }
  if (*fft_size <= 0) {
    if (!(f0_floor > 0)) { ctx->err = "cheaptrick: f0_floor must be positive"; return KWY_EINVAL; }
    *fft_size = kwy_cheaptrick_fft_size(fs, f0_floor);
  }
  return KWY_OK;
}

extern "C" int kwy_cheaptrick_dev(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                                  const double *t, const double *f0, int64_t T, double q1,
                                  double f0_floor, int fft_size, double out_div, double *out) {
  KWY_TRY(ct_check(ctx, x, x_length, fs, t, f0, T, out, &fft_size, f0_floor));
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, ct_scratch_bytes(T)));
  return cheaptrick_one(ctx, x, x_length, fs, t, f0, T, q1, fft_size, out_div, out);
}

// pyworld.cheaptrick for `count` utterances of one sampling rate in as few launches as possible (KWY_BATCH_MAX
// utterances each): one grid over all frames
extern "C" int kwy_cheaptrick_batch_dev(kwy_ctx *ctx, const kwy_utterance *utts, int count, int fs, double q1,
                                        double f0_floor, int fft_size, double out_div) {
  if (!ctx) return KWY_EINVAL;
  if (!utts || count < 1) { ctx->err = "cheaptrick_batch: bad argument"; return KWY_EINVAL; }
  size_t scratch = 0;
  int64_t frames = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_utterance &q = utts[i];
    KWY_TRY(ct_check(ctx, q.x, q.x_length, fs, q.temporal_positions, q.f0, q.f0_length, q.out, &fft_size, f0_floor));
    scratch += ct_scratch_bytes(q.f0_length);
    frames += q.f0_length;
  }
  if (frames > 0x7fffffff) { ctx->err = "cheaptrick_batch: too many frames"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, scratch));
  for (int i0 = 0; i0 < count; i0 += KWY_BATCH_MAX) {
    ct_batch b;
    b.n = count - i0 < KWY_BATCH_MAX ? count - i0 : KWY_BATCH_MAX;
    for (int u = 0; u < b.n; ++u) {
      const kwy_utterance &q = utts[i0 + u];
      b.u[u] = ct_view{q.x, q.temporal_positions, q.f0, q.out, nullptr, (int)q.x_length, (int)q.f0_length};
    }
    KWY_TRY(cheaptrick_core(ctx, b, fs, q1, fft_size, out_div));
  }
  return KWY_OK;
}

// CheapTrick + sp2mc in one: the mel-cepstra of `count` utterances (device pointers, not synchronised) for a consumer
// that does not need the envelopes themselves -- Analyzer.extract_spectrum_envelope followed by `.mel_cepstrum`
// (kwiiyatta/vocoder/world.py:43-52, vocoder/mcep.py:68-71).  utts[i].out: f0_length x (order + 1).
extern "C" int kwy_cheaptrick_mcep_batch_dev(kwy_ctx *ctx, const kwy_utterance *utts, int count, int fs, double q1,
                                             double f0_floor, int fft_size, double out_div, int order, double alpha) {
  if (!ctx) return KWY_EINVAL;
  if (!utts || count < 1 || order < 1 || order > 63 || !(fabs(alpha) < 1.0) || !(out_div > 0)) {
    ctx->err = "cheaptrick_mcep_batch: bad argument (order 1..63)";
    return KWY_EINVAL;
  }
  size_t scratch = 0;
  int64_t frames = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_utterance &q = utts[i];
    KWY_TRY(ct_check(ctx, q.x, q.x_length, fs, q.temporal_positions, q.f0, q.f0_length, q.out, &fft_size, f0_floor));
    scratch += ct_scratch_bytes(q.f0_length);
    frames += q.f0_length;
  }
  if (frames > 0x7fffffff) { ctx->err = "cheaptrick_mcep_batch: too many frames"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  // (the cepstra rows are at most K + 3 doubles; the passes of a call run one after the other but keep their own rows)
  KWY_TRY(kwy_arena_begin(ctx, scratch + ct_mcep_bytes(frames, fft_size / 2 + 4) + 4096 * ((count + KWY_BATCH_MAX - 1) / KWY_BATCH_MAX)));
  for (int i0 = 0; i0 < count; i0 += KWY_BATCH_MAX) {
    ct_batch b;
    b.n = count - i0 < KWY_BATCH_MAX ? count - i0 : KWY_BATCH_MAX;
    for (int u = 0; u < b.n; ++u) {
      const kwy_utterance &q = utts[i0 + u];
      b.u[u] = ct_view{q.x, q.temporal_positions, q.f0, q.out, nullptr, (int)q.x_length, (int)q.f0_length};
    }
    KWY_TRY(cheaptrick_core(ctx, b, fs, q1, fft_size, out_div, order, alpha));
  }
  return KWY_OK;
}

extern "C" int kwy_cheaptrick(kwy_ctx *ctx, const double *x, int64_t x_length, int fs,
                              const double *t, const double *f0, int64_t T, double q1,
                              double f0_floor, int fft_size, double out_div, double *out) {
  KWY_TRY(ct_check(ctx, x, x_length, fs, t, f0, T, out, &fft_size, f0_floor));
  KWY_HIP(hipSetDevice(ctx->device));
  const int K = fft_size / 2 + 1;
  for (int64_t i = 0; i < T; ++i)
    if (!(f0[i] < fs * 0.375)) { ctx->err = "cheaptrick: f0 must be below 3*fs/8"; return KWY_EINVAL; }
  size_t bx = kwy_pad(sizeof(double) * x_length), bt = kwy_pad(sizeof(double) * T);
  size_t bo = kwy_pad(sizeof(double) * T * K);
  KWY_TRY(kwy_arena_begin(ctx, ct_scratch_bytes(T) + bx + 2 * bt + bo));
  double *dx = kwy_arena<double>(ctx, x_length), *dt = kwy_arena<double>(ctx, T);
  double *df0 = kwy_arena<double>(ctx, T), *dout = kwy_arena<double>(ctx, (size_t)T * K);
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * x_length, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dt, t, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(df0, f0, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(cheaptrick_one(ctx, dx, x_length, fs, dt, df0, T, q1, fft_size, out_div, dout));
  KWY_HIP(hipMemcpyAsync(out, dout, sizeof(double) * T * K, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
