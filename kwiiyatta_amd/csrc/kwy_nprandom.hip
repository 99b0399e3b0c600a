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
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

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
  if (base < 0) {                     // a poisoned state (k_np_advance): no draws, the poison stays
    for (int64_t i = tid; i < n; i += KWY_THREADS) out[i] = __builtin_nan("");
    return;
  }
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

// ------------------------------------------------------------------ large requests
// The single-workgroup kernel above spends most of its time on the Box-Muller arithmetic of 156 attempts per round with
// one wavefront per SIMD.  For large requests the raw state words of as many 624-word blocks as the request can
// possibly consume are laid out first; the attempts are then judged by the whole chip: count the accepted ones per
// tile, scan the tile counts, write the outputs of the first `need` accepted attempts in order, and move the generator
// state to the word after the last one used.
//   kbuf[b][624]  block 0 = the state's current key, block b its b-th successor; attempt a takes the words
//                 pos + 4a .. pos + 4a + 3 of that stream (tempered on the fly)
//
// Laying out the words was ONE workgroup until round 4 (the recurrence z[t] = z[t - 227] ^ f(z[t - 624], z[t - 623])
// reaches back only 227 words: 0.62 ms per aligned pair, the one serial kernel of the pad spectra).  The stream is a
// linear recurrence over GF(2) with the primitive characteristic polynomial phi of degree 19937, so the window
// z[N .. N + 623] at any distance N is a fixed linear combination of the first 19937 windows:
//     z[N + k] = XOR over the set coefficients i of (x^(N-1) mod phi) of z[k + 1 + i]          (k = 0 .. 623)
// (the identity holds for every bit of every word from index 1 on; word 0 of a freshly seeded key has 31 bits that
// belong to no state).  The blocks are cut into segments of NP_SEG blocks: segment 0 runs from the state's key, the
// start key of segment s is the combination above with N = 624 NP_SEG s, read off the first 33 blocks -- one workgroup
// per segment, ~10 000 XORs per word --, and all segments then generate side by side, a workgroup each.
//   host, once per process   phi by Berlekamp-Massey on one output bit (2 x 19937 terms), h = x^(624 NP_SEG) mod phi,
//                            coefficient words of x^(624 NP_SEG s - 1) = h^s / x for the segments in use
//   k_np_words               1 workgroup: the first 33 blocks (all of them when the request is short)
//   k_np_jump                1 workgroup per segment >= 1: its start key
//   k_np_words_seg           1 workgroup per segment: its blocks
struct np_job {
  int64_t n;            // outputs wanted
  int64_t attempts;     // attempts laid out (an upper bound of those needed)
  int64_t n_each;       // outputs per destination block (outs[i / n_each][i % n_each])
  double loc, scale;
  int take_abs;
  double *const *outs;  // device array of the destination blocks
};
#define NP_TILE (KWY_THREADS * 4)     // attempts per workgroup of the counting / writing kernels

__device__ __forceinline__ void np_words_run(uint32_t *__restrict__ kbuf, int first, int last);
// the first `nblocks` blocks, one workgroup
// (A serial workgroup shares its CU with workgroups of whatever else the chip is running and runs ~2.7x
// slower beside a full load than alone; raising its wave priority with s_setprio was measured: no effect.)
__global__ __launch_bounds__(KWY_THREADS) void k_np_words(const np_state *__restrict__ st, int nblocks,
                                                         uint32_t *__restrict__ kbuf) {
  for (int i = threadIdx.x; i < MT_N; i += KWY_THREADS) kbuf[i] = st->key[i];
  __syncthreads();         // (workgroup-scope fence: the key is read back from kbuf below)
  np_words_run(kbuf, 0, nblocks);
}

#define NP_SEG 512                 // blocks per segment
#define NP_HEAD 33                 // blocks the jumps read: 33 x 624 >= 624 + 19937 words
#define NP_MAXSEG 1024             // segments a request may have (beyond: the one-workgroup layout)
#define NP_POLY_WORDS 624          // 19968 coefficient bits per jump polynomial
#define NP_JUMP_NT 1024

// kbuf block NP_SEG s <- the key of segment s = blockIdx.x + 1 (see above).  z = the first NP_HEAD blocks, staged in
// LDS; thread t of tap group q accumulates words t, t + 256, t + 512 over the q-th quarter of the polynomial's words.
__global__ __launch_bounds__(NP_JUMP_NT) void k_np_jump(uint32_t *__restrict__ kbuf, const uint32_t *__restrict__ polys) {
  extern __shared__ uint32_t zs[];                 // NP_HEAD * 624 words, the polynomial, 3 x 3 x 256 partial sums
  const int tid = threadIdx.x, t = tid & 255, q = tid >> 8;
  const int seg = blockIdx.x + 1;
  uint32_t *gs = zs + NP_HEAD * MT_N, *part = gs + NP_POLY_WORDS;
  for (int i = tid; i < NP_HEAD * MT_N; i += NP_JUMP_NT) zs[i] = kbuf[i];
  if (tid < NP_POLY_WORDS) gs[tid] = polys[(size_t)(seg - 1) * NP_POLY_WORDS + tid];
  __syncthreads();
  uint32_t a0 = 0, a1 = 0, a2 = 0;
  const uint32_t *z0 = zs + t + 1;
  const int w0 = q * (NP_POLY_WORDS / 4), w1 = w0 + NP_POLY_WORDS / 4;
  uint32_t next = gs[w0];
  for (int w = w0; w < w1; ++w) {
    uint32_t bits = __builtin_amdgcn_readfirstlane(next);
    next = gs[min(w + 1, w1 - 1)];                 // (the next word is on its way while this one's taps are applied)
    const uint32_t *zw = z0 + 32 * w;
    while (bits) {
      const int i = __builtin_ctz(bits);
      bits &= bits - 1;
      a0 ^= zw[i];
      a1 ^= zw[i + 256];
      if (t < MT_N - 512) a2 ^= zw[i + 512];
    }
  }
  if (q > 0) { part[((q - 1) * 3 + 0) * 256 + t] = a0; part[((q - 1) * 3 + 1) * 256 + t] = a1; part[((q - 1) * 3 + 2) * 256 + t] = a2; }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { a0 ^= part[(r * 3 + 0) * 256 + t]; a1 ^= part[(r * 3 + 1) * 256 + t]; a2 ^= part[(r * 3 + 2) * 256 + t]; }
    uint32_t *dst = kbuf + (size_t)seg * NP_SEG * MT_N;
    dst[t] = a0;
    dst[t + 256] = a1;
    if (t < MT_N - 512) dst[t + 512] = a2;
  }
}
#define NP_JUMP_LDS (sizeof(uint32_t) * (NP_HEAD * MT_N + NP_POLY_WORDS + 9 * 256))

// the blocks behind kbuf block `first` up to (not including) block `last`, from the key stored at `first`
__device__ __forceinline__ void np_words_run(uint32_t *__restrict__ kbuf, int first, int last) {
  __shared__ uint32_t mt[2][MT_N];
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += KWY_THREADS) mt[0][i] = kbuf[(size_t)first * MT_N + i];
  __syncthreads();
  int cur = 0;
  for (int b = first + 1; b < last; ++b) {
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
// segment s = blockIdx.x: blocks NP_SEG s .. NP_SEG (s + 1) - 1 (segment 0 continues behind the head)
__global__ __launch_bounds__(KWY_THREADS) void k_np_words_seg(uint32_t *__restrict__ kbuf, int nblocks) {
  const int s = blockIdx.x;
  const int first = s == 0 ? NP_HEAD - 1 : s * NP_SEG;
  const int last = min(nblocks, (s + 1) * NP_SEG);
  if (first + 1 < last) np_words_run(kbuf, first, last);
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
    // (pos < 0: a poisoned state -- nothing is accepted, k_np_scan flags the shortage, k_np_emit writes NaN)
    if (pos >= 0 && a < job.attempts && np_attempt(kbuf, pos, a, &x1, &x2, &r2)) ++c;
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
  if (pos < 0) {      // poisoned by an earlier shortage: the outputs of this request are NaN, not stale or garbage pads
    const int64_t stride = (int64_t)gridDim.x * KWY_THREADS;
    for (int64_t o = (int64_t)blockIdx.x * KWY_THREADS + tid; o < job.n; o += stride)
      job.outs[o / job.n_each][o % job.n_each] = __builtin_nan("");
    return;
  }
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
  // a shortage of attempts (the layout's margin is ~14 standard deviations): outputs are missing.  The state is
  // poisoned (pos = -1: no numpy state has it) so that whoever reads it back learns of it instead of a silent
  // divergence from numpy's stream.
  if (info[3] != 0 || st->pos < 0) { if (threadIdx.x == 0) { atomicExch(status, 1); st->pos = -1; } return; }
  const int64_t last = info[2];
  if (last < 0) return;                                  // nothing but the cached value was needed
  const int64_t g_last = (int64_t)st->pos + 4 * last + 3;    // stream index of the last word consumed
  const int64_t b = g_last / MT_N;
  __syncthreads();
  for (int i = threadIdx.x; i < MT_N; i += KWY_THREADS) st->key[i] = kbuf[(size_t)b * MT_N + i];
  if (threadIdx.x == 0) st->pos = (int)(g_last - b * MT_N) + 1;
}

// ------------------------------------------------------------------ host: MT19937's jump polynomials
namespace {
constexpr int MTD = 19937;                 // degree of the characteristic polynomial
constexpr int PW = 313;                    // 64-bit words of a polynomial of degree <= 19937 (and some)
typedef std::vector<uint64_t> poly;        // coefficient i = bit i

inline int pbit(const poly &p, int i) { return (int)((p[i >> 6] >> (i & 63)) & 1u); }
// r ^= a << sh  (r long enough)
inline void pxor_shift(poly &r, const poly &a, int na, int sh) {
  const int ws = sh >> 6, bs = sh & 63;
  if (bs == 0) { for (int q = 0; q < na; ++q) r[q + ws] ^= a[q]; return; }
  for (int q = 0; q < na; ++q) {
    const uint64_t v = a[q];
    if (!v) continue;
    r[q + ws] ^= v << bs;
    r[q + ws + 1] ^= v >> (64 - bs);
  }
}
struct MtJump {
  std::mutex mu;
  bool ready = false, failed = false;
  poly phi;                         // characteristic polynomial, PW words
  poly h;                           // x^(624 NP_SEG) mod phi
  std::vector<poly> g;              // g[s - 1] = x^(624 NP_SEG s - 1) mod phi
  poly hs;                          // h^(number of polynomials made)
  void reduce(poly &r) const {      // r: 2 PW + 1 words -> degree < MTD
    for (int i = 2 * 64 * PW - 1; i >= MTD; --i)
      if (pbit(r, i)) pxor_shift(r, phi, PW, i - MTD);
  }
  poly mulmod(const poly &a, const poly &b) const {
    poly r(2 * PW + 2, 0);
    for (int q = 0; q < PW; ++q) {
      uint64_t bits = a[q];
      while (bits) {
        const int i = __builtin_ctzll(bits) + 64 * q;
        bits &= bits - 1;
        pxor_shift(r, b, PW, i);
      }
    }
    reduce(r);
    r.resize(PW);
    return r;
  }
  // a / x mod phi (phi's constant term is 1: it is irreducible)
  poly divx(const poly &a) const {
    poly r = a;
    if (r[0] & 1u) for (int q = 0; q < PW; ++q) r[q] ^= phi[q];
    for (int q = 0; q < PW; ++q) r[q] = (r[q] >> 1) | (q + 1 < PW ? r[q + 1] << 63 : 0);
    return r;
  }
  static void host_words(std::vector<uint32_t> &z, size_t n) {      // z: 624 words in, n words out
    z.resize(n);
    for (size_t t = MT_N; t < n; ++t) {
      const uint32_t y = (z[t - MT_N] & 0x80000000u) | (z[t - MT_N + 1] & 0x7fffffffu);
      z[t] = z[t - MT_N + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
  bool init(std::string &err) {
    if (ready) return true;
    if (failed) { err = "np_normal: MT19937 jump tables failed their self-check"; return false; }
    // any stream will do: init_genrand(5489)
    std::vector<uint32_t> z(MT_N);
    z[0] = 5489u;
    for (int i = 1; i < MT_N; ++i) z[i] = 1812433253u * (z[i - 1] ^ (z[i - 1] >> 30)) + (uint32_t)i;
    const int nseq = 2 * MTD + 64;
    host_words(z, (size_t)nseq + MT_N + 64 * MT_N);
    // Berlekamp-Massey on a_t = bit 0 of z[t + 1].  C, B: connection polynomials; R: the sequence seen so far,
    // newest term in bit 0 (the discrepancy is the parity of C & R).
    poly C(PW + 1, 0), B(PW + 1, 0), R(PW + 1, 0), T;
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int i = 0; i < nseq; ++i) {
      for (int q = PW; q > 0; --q) R[q] = (R[q] << 1) | (R[q - 1] >> 63);
      R[0] = (R[0] << 1) | (uint64_t)(z[i + 1] & 1u);
      uint64_t acc = 0;
      for (int q = 0; q <= PW; ++q) acc ^= C[q] & R[q];
      if ((__builtin_popcountll(acc) & 1) == 0) { ++m; continue; }
      if (2 * L <= i) {
        T = C;
        if (m <= 64 * PW) pxor_shift(C, B, PW - (m >> 6), m);
        L = i + 1 - L; B = T; m = 1;
      } else {
        if (m <= 64 * PW) pxor_shift(C, B, PW - (m >> 6), m);
        ++m;
      }
    }
    if (L != MTD) { failed = true; err = "np_normal: MT19937: unexpected linear complexity"; return false; }
    phi.assign(PW, 0);
    for (int j = 0; j <= MTD; ++j)
      if (pbit(C, j)) phi[(MTD - j) >> 6] ^= (uint64_t)1 << ((MTD - j) & 63);    // phi(x) = x^L C(1/x)
    // h = x^(624 NP_SEG) mod phi
    {
      poly result(PW, 0), base(PW, 0);
      result[0] = 1; base[0] = 2;
      for (uint64_t n = (uint64_t)MT_N * NP_SEG; n; n >>= 1) {
        if (n & 1) result = mulmod(result, base);
        if (n > 1) base = mulmod(base, base);
      }
      h = result;
    }
    hs.assign(PW, 0);
    hs[0] = 1;
    // self-check on the host stream: the key of segment 1 from the first NP_HEAD blocks
    {
      extend(1);
      std::vector<uint32_t> w(MT_N);
      w[0] = 19650218u;
      for (int i = 1; i < MT_N; ++i) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + (uint32_t)i;
      host_words(w, (size_t)MT_N * NP_SEG + MT_N);
      const poly &g1 = g[0];
      for (int k : {0, 1, 227, 623}) {
        uint32_t acc = 0;
        for (int i = 0; i < MTD; ++i) if (pbit(g1, i)) acc ^= w[(size_t)k + 1 + i];
        if (acc != w[(size_t)MT_N * NP_SEG + k]) { failed = true; err = "np_normal: MT19937 jump tables failed their self-check"; return false; }
      }
    }
    ready = true;
    return true;
  }
  void extend(int nseg) {           // polynomials of segments 1 .. nseg
    while ((int)g.size() < nseg) {
      hs = mulmod(hs, h);
      g.push_back(divx(hs));
    }
  }
} g_mt;
}  // namespace

// the device table of the jump polynomials of segments 1 .. nseg (NP_MAXSEG rows allocated once per context: the
// pointer stays valid for captured graphs; rows are filled as requests need them)
static int np_get_polys(kwy_ctx *ctx, int nseg, const uint32_t **out) {
  std::lock_guard<std::mutex> lock(g_mt.mu);
  if (!g_mt.init(ctx->err)) return KWY_EHIP;
  g_mt.extend(nseg);
  auto it = ctx->d_mats.find("mtjump");
  if (it == ctx->d_mats.end()) {
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(uint32_t) * (size_t)NP_MAXSEG * NP_POLY_WORDS));
    it = ctx->d_mats.emplace("mtjump", d).first;
    ctx->i_vals["mtjump"] = 0;
  }
  uint32_t *d = (uint32_t *)it->second;
  int64_t &have = ctx->i_vals["mtjump"];
  if (have < nseg) {
    std::vector<uint32_t> hbuf((size_t)(nseg - have) * NP_POLY_WORDS, 0u);
    for (int64_t sgi = have; sgi < nseg; ++sgi)
      memcpy(hbuf.data() + (size_t)(sgi - have) * NP_POLY_WORDS, g_mt.g[(size_t)sgi].data(), sizeof(uint32_t) * NP_POLY_WORDS);
    KWY_HIP(hipMemcpy(d + (size_t)have * NP_POLY_WORDS, hbuf.data(), sizeof(uint32_t) * hbuf.size(), hipMemcpyHostToDevice));
    have = nseg;
  }
  *out = d;
  return KWY_OK;
}

// Not part of the ABI (include/kwy.h): the host half of the jump tables alone -- characteristic polynomial, the
// polynomials of `nseg` segments and the self-check against a directly generated stream -- for the CPU test suite.
// words: optional nseg x 624 uint32 output.  Returns 0 when the tables passed.
extern "C" int kwy_np_jump_tables_host(int nseg, uint32_t *words) {
  std::lock_guard<std::mutex> lock(g_mt.mu);
  std::string err;
  if (nseg < 1 || nseg > NP_MAXSEG || !g_mt.init(err)) return -1;
  g_mt.extend(nseg);
  if (words)
    for (int sgi = 0; sgi < nseg; ++sgi) memcpy(words + (size_t)sgi * NP_POLY_WORDS, g_mt.g[(size_t)sgi].data(), sizeof(uint32_t) * NP_POLY_WORDS);
  return 0;
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

// the destination list travels in kernel arguments (a copy node from a caller's host array would not survive in a
// captured graph): NP_PTRS pointers per launch
#define NP_PTRS 128
struct np_ptrs { int n, first; double *p[NP_PTRS]; };
__global__ void k_np_set_outs(np_ptrs a, double **__restrict__ douts) {
  if ((int)threadIdx.x < a.n) douts[a.first + threadIdx.x] = a.p[threadIdx.x];
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
  for (int c0 = 0; c0 < count; c0 += NP_PTRS) {
    np_ptrs a;
    a.n = std::min(NP_PTRS, count - c0); a.first = c0;
    for (int c = 0; c < NP_PTRS; ++c) a.p[c] = outs[c0 + (c < a.n ? c : 0)];
    hipLaunchKernelGGL(k_np_set_outs, dim3(1), dim3(NP_PTRS), 0, ctx->stream, a, douts);
  }
  job.outs = douts;
  KWY_HIP(hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
  const int nseg = (nblocks + NP_SEG - 1) / NP_SEG;
  if (nseg < 2 || nseg > NP_MAXSEG) {
    KWY_PROF(ctx, "k_np_words", hipLaunchKernelGGL(k_np_words, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, nblocks, kbuf));
  } else {
    const uint32_t *polys;
    KWY_TRY(np_get_polys(ctx, nseg - 1, &polys));
    KWY_HIP(hipFuncSetAttribute((const void *)k_np_jump, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NP_JUMP_LDS));
    KWY_PROF(ctx, "k_np_words", hipLaunchKernelGGL(k_np_words, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, state, NP_HEAD, kbuf));
    KWY_PROF(ctx, "k_np_jump", hipLaunchKernelGGL(k_np_jump, dim3(nseg - 1), dim3(NP_JUMP_NT), NP_JUMP_LDS, ctx->stream, kbuf, polys));
    KWY_PROF(ctx, "k_np_words_seg", hipLaunchKernelGGL(k_np_words_seg, dim3(nseg), dim3(KWY_THREADS), 0, ctx->stream, kbuf, nblocks));
  }
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
