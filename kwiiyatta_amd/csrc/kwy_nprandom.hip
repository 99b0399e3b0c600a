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

// ------------------------------------------------------------------ large requests: five launches
// The single-workgroup kernel above spends most of its time on the Box-Muller arithmetic of 156 attempts per round with
// one wavefront per SIMD.  For large requests only the MT19937 recurrence stays serial (one workgroup writes the raw
// state words of as many 624-word blocks as the request can possibly consume); the attempts are then judged by the
// whole chip: count the accepted ones per tile, scan the tile counts, write the outputs of the first `need` accepted
// attempts in order, and move the generator state to the word after the last one used.
//   kbuf[b][624]  block 0 = the state's current key, block b its b-th successor; attempt a takes the words
//                 pos + 4a .. pos + 4a + 3 of that stream (tempered on the fly)
struct np_job {
  int64_t n;            // outputs wanted
  int64_t attempts;     // attempts laid out (an upper bound of those needed)
  int64_t n_each;       // outputs per destination block (outs[i / n_each][i % n_each])
  double loc, scale;
  int take_abs;
  double *const *outs;  // device array of the destination blocks
};
#define NP_TILE (KWY_THREADS * 4)     // attempts per workgroup of the counting / writing kernels

__global__ __launch_bounds__(KWY_THREADS) void k_np_words(const np_state *__restrict__ st, int nblocks,
                                                         uint32_t *__restrict__ kbuf) {
  __shared__ uint32_t mt[2][MT_N];
  const int tid = threadIdx.x;
  // (This one serial workgroup shares its CU with workgroups of whatever else the chip is running and runs ~2.7x
  // slower beside a full load than alone; raising its wave priority with s_setprio was measured: no effect.)
  for (int i = tid; i < MT_N; i += KWY_THREADS) { const uint32_t v = st->key[i]; mt[0][i] = v; kbuf[i] = v; }
  __syncthreads();
  int cur = 0;
  for (int b = 1; b < nblocks; ++b) {
    const uint32_t *o = mt[cur];
    uint32_t *w = mt[cur ^ 1];
    uint32_t *dst = kbuf + (size_t)b * MT_N;
    if (tid < MT_N - MT_M) { const uint32_t v = mt_twist(o[tid], o[tid + 1], o[tid + MT_M]); w[tid] = v; dst[tid] = v; }
    kwy_lds_barrier();      // (the block's words stream out to global memory meanwhile: nothing here reads them)
    if (tid < MT_N - MT_M) {
      const int k = (MT_N - MT_M) + tid;
      const uint32_t v = mt_twist(o[k], o[k + 1], w[k - (MT_N - MT_M)]);
      w[k] = v; dst[k] = v;
    }
    kwy_lds_barrier();
    {
      const int k = 2 * (MT_N - MT_M) + tid;
      if (k < MT_N) {
        const uint32_t v = k < MT_N - 1 ? mt_twist(o[k], o[k + 1], w[k - (MT_N - MT_M)]) : mt_twist(o[k], w[0], w[MT_M - 1]);
        w[k] = v; dst[k] = v;
      }
    }
    kwy_lds_barrier();
    cur ^= 1;
  }
}

__device__ __forceinline__ bool np_attempt(const uint32_t *__restrict__ kbuf, int pos, int64_t a, double *x1, double *x2,
                                           double *r2) {
  const int64_t g = (int64_t)pos + 4 * a;
  const uint32_t wa = mt_temper(kbuf[g]), wb = mt_temper(kbuf[g + 1]), wc = mt_temper(kbuf[g + 2]), wd = mt_temper(kbuf[g + 3]);
  *x1 = 2.0 * np_uniform53(wa, wb) - 1.0;
  *x2 = 2.0 * np_uniform53(wc, wd) - 1.0;
  *r2 = *x1 * *x1 + *x2 * *x2;
  return !(*r2 >= 1.0 || *r2 == 0.0);
}

__global__ __launch_bounds__(KWY_THREADS) void k_np_count(const np_state *__restrict__ st, const uint32_t *__restrict__ kbuf,
                                                         np_job job, int *__restrict__ counts) {
  __shared__ int red[KWY_WAVES];
  const int pos = st->pos;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t a = (int64_t)blockIdx.x * NP_TILE + j * KWY_THREADS + threadIdx.x;
    double x1, x2, r2;
    if (a < job.attempts && np_attempt(kbuf, pos, a, &x1, &x2, &r2)) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the tile counts; emits the cached Gaussian of the previous call first; flags a shortage
__global__ __launch_bounds__(KWY_THREADS) void k_np_scan(np_state *__restrict__ st, np_job job, int ntiles,
                                                        int *__restrict__ counts, int64_t *__restrict__ info) {
  __shared__ uint64_t tot[KWY_THREADS];
  __shared__ int64_t s_total;
  const int had = st->has_gauss;
  kwy_block_count_scan<KWY_THREADS>([&](int64_t i) -> uint64_t { return (uint64_t)counts[i]; }, ntiles,
                                    (uint64_t *)(info + 8), tot);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int64_t first = had ? 1 : 0;                 // outputs served by the cached value
    if (had) {
      const double v = job.loc + job.scale * st->gauss;
      job.outs[0][0] = job.take_abs ? fabs(v) : v;
      st->has_gauss = 0;
      st->gauss = 0.0;
    }
    const int64_t need = (job.n - first + 1) / 2;      // accepted attempts wanted
    info[0] = first;
    info[1] = need;
    info[2] = -1;                                      // the last attempt used (set by k_np_emit)
    info[3] = (int64_t)(info + 8)[ntiles] < need ? 1 : 0;   // shortage: cannot happen with the margin laid out
  }
}

__global__ __launch_bounds__(KWY_THREADS) void k_np_emit(np_state *__restrict__ st, const uint32_t *__restrict__ kbuf,
                                                        np_job job, int64_t *__restrict__ info) {
  __shared__ int wsum[4][KWY_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pos = st->pos;
  const int64_t first = info[0], need = info[1];
  const int64_t tile_base = (int64_t)((const uint64_t *)(info + 8))[blockIdx.x];
  if (tile_base >= need) return;
  double x1[4], x2[4], r2[4];
  bool ok[4];
  unsigned long long bal[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t a = (int64_t)blockIdx.x * NP_TILE + j * KWY_THREADS + tid;
    ok[j] = a < job.attempts && np_attempt(kbuf, pos, a, &x1[j], &x2[j], &r2[j]);
    bal[j] = __ballot(ok[j]);
    if (lane == 0) wsum[j][wv] = __popcll(bal[j]);
  }
  __syncthreads();
  int64_t before = tile_base;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t rank = before + __popcll(bal[j] & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < KWY_WAVES; ++w) {
      if (w < wv) rank += wsum[j][w];
      before += wsum[j][w];
    }
    if (ok[j] && rank < need) {
      const double f = sqrt(-2.0 * log(r2[j]) / r2[j]);
      const int64_t o = first + 2 * rank;
      const double va = job.loc + job.scale * (f * x2[j]), vb = job.loc + job.scale * (f * x1[j]);
      job.outs[o / job.n_each][o % job.n_each] = job.take_abs ? fabs(va) : va;
      if (o + 1 < job.n) job.outs[(o + 1) / job.n_each][(o + 1) % job.n_each] = job.take_abs ? fabs(vb) : vb;
      else { st->gauss = f * x1[j]; st->has_gauss = 1; }
      if (rank == need - 1) info[2] = (int64_t)blockIdx.x * NP_TILE + j * KWY_THREADS + tid;
    }
  }
}

// the state moves to the word after the last attempt used
__global__ __launch_bounds__(KWY_THREADS) void k_np_advance(np_state *__restrict__ st, const uint32_t *__restrict__ kbuf,
                                                           const int64_t *__restrict__ info, int *__restrict__ status) {
  if (info[3] != 0) { if (threadIdx.x == 0) atomicExch(status, 1); return; }
  const int64_t last = info[2];
  if (last < 0) return;                                  // nothing but the cached value was needed
  const int64_t g_last = (int64_t)st->pos + 4 * last + 3;    // stream index of the last word consumed
  const int64_t b = g_last / MT_N;
  __syncthreads();
  for (int i = threadIdx.x; i < MT_N; i += KWY_THREADS) st->key[i] = kbuf[(size_t)b * MT_N + i];
  if (threadIdx.x == 0) st->pos = (int)(g_last - b * MT_N) + 1;
}

// ------------------------------------------------------------------ C ABI
extern "C" int64_t kwy_np_state_bytes(void) { return (int64_t)sizeof(np_state); }

static int np_check(kwy_ctx *ctx, const void *state, int64_t n, const void *out) {
  if (!ctx) return KWY_EINVAL;
  if (!state || n < 0 || (n > 0 && !out)) { ctx->err = "np_normal: bad argument"; return KWY_EINVAL; }
  return KWY_OK;
}

#define NP_SMALL 4096     // up to here the single-workgroup kernel (no scratch, exact for any acceptance pattern)

static int64_t np_attempts_cap(int64_t n) { return (int64_t)((n / 2 + 1) * 1.35) + 256; }

static size_t np_scratch_bytes(int64_t n, int count = 1) {
  if (n <= NP_SMALL) return 256;
  const int64_t attempts = np_attempts_cap(n);
  const int64_t nblocks = 2 + (MT_N + 4 * attempts) / MT_N;
  const int64_t ntiles = (attempts + NP_TILE - 1) / NP_TILE;
  return kwy_pad(sizeof(uint32_t) * (size_t)nblocks * MT_N) + kwy_pad(sizeof(int) * ntiles) +
         kwy_pad(sizeof(int64_t) * (size_t)(ntiles + 16)) + kwy_pad(64) + kwy_pad(sizeof(double *) * (size_t)count);
}

// outs: HOST array of `count` device pointers to blocks of n_each doubles each, filled in order from one continuous
// stream
static int np_core(kwy_ctx *ctx, np_state *state, double loc, double scale, int take_abs, int count, int64_t n_each,
                   double *const *outs) {
  const int64_t n = (int64_t)count * n_each;
  if (n <= NP_SMALL) {
    for (int c = 0; c < count; ++c)
      KWY_PROF(ctx, "k_np_normal", hipLaunchKernelGGL(k_np_normal, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, loc,
                                                      scale, take_abs, n_each, outs[c]));
    KWY_HIP(hipGetLastError());
    return KWY_OK;
  }
  np_job job;
  job.n = n; job.n_each = n_each; job.loc = loc; job.scale = scale; job.take_abs = take_abs;
  job.attempts = np_attempts_cap(n);
  if (2 + (MT_N + 4 * job.attempts) / MT_N > 0x7fffffff) { ctx->err = "np_normal: request too large"; return KWY_EINVAL; }
  const int nblocks = (int)(2 + (MT_N + 4 * job.attempts) / MT_N);
  const int ntiles = (int)((job.attempts + NP_TILE - 1) / NP_TILE);
  uint32_t *kbuf = kwy_arena<uint32_t>(ctx, (size_t)nblocks * MT_N);
  int *counts = kwy_arena<int>(ctx, ntiles);
  int64_t *info = kwy_arena<int64_t>(ctx, (size_t)ntiles + 16);
  int *status = kwy_arena<int>(ctx, 16);
  double **douts = kwy_arena<double *>(ctx, (size_t)count);
  if (!kbuf || !counts || !info || !status || !douts) { ctx->err = "np_normal: scratch arena too small"; return KWY_ENOMEM; }
  // (the pointer list is small and comes from pageable host memory: the copy is done when the call returns)
  KWY_HIP(hipMemcpyAsync(douts, outs, sizeof(double *) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  job.outs = douts;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
  KWY_PROF(ctx, "k_np_words", hipLaunchKernelGGL(k_np_words, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, nblocks, kbuf));
  hipLaunchKernelGGL(k_np_count, dim3(ntiles), dim3(KWY_THREADS), 0, ctx->stream, state, kbuf, job, counts);
  hipLaunchKernelGGL(k_np_scan, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, job, ntiles, counts, info);
  KWY_PROF(ctx, "k_np_emit", hipLaunchKernelGGL(k_np_emit, dim3(ntiles), dim3(KWY_THREADS), 0, ctx->stream, state, kbuf, job, info));
  hipLaunchKernelGGL(k_np_advance, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, kbuf, info, status);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_np_normal_dev(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n,
                                 double *out) {
  KWY_TRY(np_check(ctx, state, n, out));
  if (n == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, np_scratch_bytes(n)));
  double *outs[1] = {out};
  return np_core(ctx, (np_state *)state, loc, scale, take_abs, 1, n, outs);
}

// `count` blocks of n_each values each from one continuous stream: what pad_silence draws for one aligned pair
// (count = 4), or for all pairs of a batch in pair order (one serial word kernel instead of one per pair)
extern "C" int kwy_np_normal_blocks_dev(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int count,
                                        int64_t n_each, double *const *outs) {
  if (!ctx) return KWY_EINVAL;
  if (!state || !outs || count < 1 || n_each < 1) { ctx->err = "np_normal_blocks: bad argument"; return KWY_EINVAL; }
  for (int c = 0; c < count; ++c)
    if (!outs[c]) { ctx->err = "np_normal_blocks: null destination"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, np_scratch_bytes((int64_t)count * n_each, count)));
  return np_core(ctx, (np_state *)state, loc, scale, take_abs, count, n_each, outs);
}

extern "C" int kwy_np_normal(kwy_ctx *ctx, void *state, double loc, double scale, int take_abs, int64_t n,
                             double *out) {
  KWY_TRY(np_check(ctx, state, n, out));
  if (n == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, np_scratch_bytes(n) + kwy_pad(sizeof(np_state)) + kwy_pad(sizeof(double) * (size_t)n)));
  np_state *ds = (np_state *)kwy_arena_alloc(ctx, sizeof(np_state));
  double *dout = kwy_arena<double>(ctx, (size_t)n);
  KWY_HIP(hipMemcpyAsync(ds, state, sizeof(np_state), hipMemcpyHostToDevice, ctx->stream));
  double *outs[1] = {dout};
  KWY_TRY(np_core(ctx, ds, loc, scale, take_abs, 1, n, outs));
  KWY_HIP(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipMemcpyAsync(state, ds, sizeof(np_state), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
