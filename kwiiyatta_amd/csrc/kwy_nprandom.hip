// kwy_nprandom.hip -- numpy's LEGACY normal generator on gfx950, draw for draw.
//
// The reference pads every utterance it aligns with "silent" spectra |N(0, EPS / fs)| taken from numpy's global
// generator: WorldSynthesizer._silence_spectrum_envelope, kwiiyatta/vocoder/world.py:158-161, called from
// pad_silence (vocoder/abc/feature.py:19-41) by align / align_even for BOTH sides of every pair -- 4 x 100 x 1025
// draws per pair at 48 kHz.  np.random.normal is serial (4.6 ms per pair on one host core: four times the GPU time
// of the whole pair), so a batch driver that wants the reference's numbers has to reproduce the generator on the
// device:
//
//   RandomState.normal      loc + scale * legacy_gauss()
//   legacy_gauss            polar Box-Muller: x1, x2 = 2 u - 1 from two 53-bit uniforms each; rejected unless
//                           0 < r2 = x1^2 + x2^2 < 1; f = sqrt(-2 log(r2) / r2); returns f x2 and keeps f x1 for
//                           the next call (has_gauss)
//   53-bit uniform          (a >> 5) * 2^26 + (b >> 6) over 2^53 from two MT19937 outputs
//
// so one ATTEMPT consumes exactly four 32-bit words and yields zero or two normals.  MT19937 advances in blocks of
// 624 words; inside a block the recurrence new[k] = new/old[k + 397 mod 624] ^ f(old[k], old[k+1]) has three
// fully parallel phases (227 + 227 + 170 words).  One workgroup runs the stream: per block 3 generation phases,
// tempering, up to 156 attempts in parallel, an ordered compaction of the accepted ones (ballot + popcount), and
// the Box-Muller arithmetic -- about 1 ms per pair, off the host and overlapping with the analysis kernels of
// other streams.  The accept pattern is bit-exact (integer -> double conversions, one multiply-add chain compiled
// without contraction); values differ from numpy's only where the device's log() rounds differently from glibc's
// (<= 1 ulp).  The generator state (key[624], pos, has_gauss, gauss: numpy's get_state() tuple) lives in device
// memory between calls and can be copied from / to numpy.
#include <math.h>

#include "kwy_internal.hpp"

#define MT_N 624
#define MT_M 397

struct np_state {
  uint32_t key[MT_N];
  int32_t pos;          // next word of key[] to be consumed (624: regenerate first)
  int32_t has_gauss;
  double gauss;
};

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v, uint32_t far) {
  const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

__device__ __forceinline__ double np_uniform53(uint32_t a, uint32_t b) {
  return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0;
}

// out[i] = loc + scale * gauss_i (absolute value when take_abs), i < n, continuing the stream of *st; *st is updated.
__global__ __launch_bounds__(KWY_THREADS) void k_np_normal(np_state *__restrict__ st, double loc, double scale,
                                                          int take_abs, int64_t n, double *__restrict__ out) {
  __shared__ uint32_t mt[2][MT_N];
  __shared__ uint32_t words[MT_N + 8];
  __shared__ int wcnt[KWY_WAVES];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto emit = [&](double g) { const double v = loc + scale * g; return take_abs ? fabs(v) : v; };

  for (int i = tid; i < MT_N; i += KWY_THREADS) mt[0][i] = st->key[i];
  int cur = 0;
  int base = st->pos;                 // index in the current block of words[r]
  int r = 0;                          // words at the front of words[] that belong to the block before
  int64_t produced = 0;
  if (n <= 0) return;
  if (st->has_gauss) {
    if (tid == 0) out[0] = emit(st->gauss);
    produced = 1;
  }
  __syncthreads();
  if (tid == 0 && produced) { st->has_gauss = 0; st->gauss = 0.0; }
  int navail = MT_N - base;
  for (int i = tid; i < navail; i += KWY_THREADS) words[i] = mt_temper(mt[0][base + i]);
  __syncthreads();

  while (produced < n) {               // uniform: every thread carries the same counters
    const int na = navail >> 2;
    if (na == 0) {
      // the next block: three parallel phases of the recurrence, old -> new array
      const uint32_t *o = mt[cur];
      uint32_t *w = mt[cur ^ 1];
      if (tid < MT_N - MT_M) w[tid] = mt_twist(o[tid], o[tid + 1], o[tid + MT_M]);
      __syncthreads();
      if (tid < MT_N - MT_M) {
        const int k = (MT_N - MT_M) + tid;
        w[k] = mt_twist(o[k], o[k + 1], w[k - (MT_N - MT_M)]);
      }
      __syncthreads();
      {
        const int k = 2 * (MT_N - MT_M) + tid;
        if (k < MT_N - 1) w[k] = mt_twist(o[k], o[k + 1], w[k - (MT_N - MT_M)]);
        else if (k == MT_N - 1) w[k] = mt_twist(o[k], w[0], w[MT_M - 1]);
      }
      __syncthreads();
      cur ^= 1;
      for (int i = tid; i < MT_N; i += KWY_THREADS) words[navail + i] = mt_temper(w[i]);
      r = navail;
      base = 0;
      navail += MT_N;
      __syncthreads();
      continue;
    }
    // ---- na (<= 157) attempts side by side: thread t takes words 4t .. 4t+3
    double x1 = 0.0, x2 = 0.0, r2 = 0.0;
    bool ok = false;
    if (tid < na) {
      const uint32_t a = words[4 * tid], b = words[4 * tid + 1], c = words[4 * tid + 2], d = words[4 * tid + 3];
      x1 = 2.0 * np_uniform53(a, b) - 1.0;
      x2 = 2.0 * np_uniform53(c, d) - 1.0;
      r2 = x1 * x1 + x2 * x2;
      ok = !(r2 >= 1.0 || r2 == 0.0);
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wcnt[wv] = __popcll(bal);
    if (tid == 0) s_last = -1;
    __syncthreads();
    int rank = __popcll(bal & ((1ull << lane) - 1ull));
    int accepted = 0;
#pragma unroll
    for (int w = 0; w < KWY_WAVES; ++w) {
      if (w < wv) rank += wcnt[w];
      accepted += wcnt[w];
    }
    const int64_t need = (n - produced + 1) / 2;        // accepted attempts still wanted
    const int use = (int64_t)accepted < need ? accepted : (int)need;
    if (ok && rank < use) {
      const double f = sqrt(-2.0 * log(r2) / r2);
      const int64_t o = produced + 2 * (int64_t)rank;
      out[o] = emit(f * x2);
      if (o + 1 < n) out[o + 1] = emit(f * x1);
      else { st->gauss = f * x1; st->has_gauss = 1; }      // odd request: the twin waits for the next call
      if (rank == use - 1) s_last = tid;
    }
    __syncthreads();
    produced += 2 * (int64_t)use;
    if ((int64_t)accepted >= need) {
      // done inside this batch: the words after the last used attempt stay unconsumed in the current block
      const int consumed = 4 * (s_last + 1);
      for (int i = tid; i < MT_N; i += KWY_THREADS) st->key[i] = mt[cur][i];
      if (tid == 0) st->pos = base + consumed - r;
      return;
    }
    // all na attempts are spent: fewer than four words stay, at the front
    const int consumed = 4 * na, left = navail - consumed;
    uint32_t keep = 0;
    if (tid < left) keep = words[consumed + tid];
    __syncthreads();
    if (tid < left) words[tid] = keep;
    // (the block they came from is finished: the generation step above sets r = left, base = 0)
    navail = left;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ C ABI
extern "C" int64_t kwy_np_state_bytes(void) { return (int64_t)sizeof(np_state); }

static int np_check(kwy_ctx *ctx, const void *state, int64_t n, const void *out) {
  if (!ctx) return KWY_EINVAL;
  if (!state || n < 0 || (n > 0 && !out)) { ctx->err = "np_normal: bad argument"; return KWY_EINVAL; }
  return KWY_OK;
}

extern "C" int kwy_np_normal_dev(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n,
                                 double *out) {
  KWY_TRY(np_check(ctx, state, n, out));
  if (n == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_PROF(ctx, "k_np_normal", hipLaunchKernelGGL(k_np_normal, dim3(1), dim3(KWY_THREADS), 0, ctx->stream,
                                                  (np_state *)state, loc, scale, take_abs, n, out));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_np_normal(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n,
                             double *out) {
  KWY_TRY(np_check(ctx, state, n, out));
  if (n == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(np_state)) + kwy_pad(sizeof(double) * (size_t)n)));
  np_state *ds = (np_state *)kwy_arena_alloc(ctx, sizeof(np_state));
  double *dout = kwy_arena<double>(ctx, (size_t)n);
  KWY_HIP(hipMemcpyAsync(ds, state, sizeof(np_state), hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_np_normal_dev(ctx, ds, loc, scale, take_abs, n, dout));
  KWY_HIP(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(state, ds, sizeof(np_state), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
