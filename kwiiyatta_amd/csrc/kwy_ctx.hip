// kwy_ctx.hip -- context, scratch arena and constant tables of libkwy.so, plus the
// two small kernels every noise-consuming stage shares (offset scan and
// generator jump-ahead).
//
// WORLD draws its analysis/synthesis noise from ONE serial xorshift128 stream
// (reference call sites: pyworld.cheaptrick / d4c / synthesize,
// kwiiyatta/vocoder/world.py:45,55,86).  To reproduce that stream without
// serialising the GPU, the transition T of the generator is treated as a
// 128x128 matrix over GF(2):
//   * T^(2^k), k < 64, lets one wavefront jump a state by any 64-bit step count
//   * with P(x) the characteristic polynomial of T, x^n mod P gives
//     T^n S0 = sum_i c_i T^i S0, so the 256 threads of a block derive their own
//     start states from 131 consecutive words of the block's stream.
#include <math.h>
#include <string.h>

#include "kwy_internal.hpp"

// ================================================================ GF(2) host maths
namespace {

struct St { uint32_t v[4]; };

inline void host_step(St &s) {
  uint32_t t = s.v[0] ^ (s.v[0] << 11);
  s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3];
  s.v[3] = (s.v[3] ^ (s.v[3] >> 19)) ^ (t ^ (t >> 8));
}

struct Mat { St col[128]; };

inline St mat_apply(const Mat &m, const St &s) {
  St o = {{0, 0, 0, 0}};
  for (int b = 0; b < 128; ++b)
    if ((s.v[b >> 5] >> (b & 31)) & 1u)
      for (int w = 0; w < 4; ++w) o.v[w] ^= m.col[b].v[w];
  return o;
}

typedef struct { uint64_t w[4]; } Poly;  // up to 256 coefficient bits

inline int poly_bit(const Poly &p, int i) { return (p.w[i >> 6] >> (i & 63)) & 1u; }
inline void poly_flip(Poly &p, int i) { p.w[i >> 6] ^= (uint64_t)1 << (i & 63); }

// Berlekamp-Massey over GF(2): shortest LFSR for s[0..n)
int berlekamp_massey(const std::vector<int> &s, std::vector<int> &C) {
  int n = (int)s.size();
  std::vector<int> B(n + 1, 0), Cc(n + 1, 0);
  B[0] = Cc[0] = 1;
  int L = 0, m = 1;
  for (int i = 0; i < n; ++i) {
    int d = s[i];
    for (int j = 1; j <= L; ++j) d ^= Cc[j] & s[i - j];
    if (d == 0) { ++m; continue; }
    std::vector<int> T = Cc;
    for (int j = 0; j + m <= n; ++j) Cc[j + m] ^= B[j];
    if (2 * L <= i) { L = i + 1 - L; B = T; m = 1; } else { ++m; }
  }
  C = Cc;
  return L;
}

Poly g_P;  // characteristic polynomial of T (degree 128)

Poly poly_mulmod(const Poly &a, const Poly &b) {
  Poly r = {{0, 0, 0, 0}};
  for (int w = 0; w < 2; ++w) {
    uint64_t bits = a.w[w];
    while (bits) {
      const int i = __builtin_ctzll(bits) + 64 * w;
      bits &= bits - 1;
      const int ws = i >> 6, bs = i & 63;  // r ^= b << i
      for (int q = 0; q < 2; ++q) {
        const uint64_t v = b.w[q];
        if (!v) continue;
        r.w[q + ws] ^= v << bs;
        if (bs && q + ws + 1 < 4) r.w[q + ws + 1] ^= v >> (64 - bs);
      }
    }
  }
  // reduce modulo P (degree 128): P = x^128 + (g_P.w[1], g_P.w[0])
  for (int i = 254; i >= 128; --i) {
    if (!poly_bit(r, i)) continue;
    const int sh = i - 128, ws = sh >> 6, bs = sh & 63;
    for (int q = 0; q < 3; ++q) {
      const uint64_t v = g_P.w[q];
      if (!v) continue;
      r.w[q + ws] ^= v << bs;
      if (bs && q + ws + 1 < 4) r.w[q + ws + 1] ^= v >> (64 - bs);
    }
  }
  return r;
}

Poly poly_xpow(uint64_t n) {  // x^n mod P
  Poly result = {{1, 0, 0, 0}};
  Poly base = {{2, 0, 0, 0}};
  while (n) {
    if (n & 1) result = poly_mulmod(result, base);
    base = poly_mulmod(base, base);
    n >>= 1;
  }
  return result;
}

bool g_tables_ready = false;
std::vector<Mat> g_pow2;  // T^(2^k)

bool build_host_tables(std::string &err) {
  if (g_tables_ready) return true;
  g_pow2.resize(64);
  for (int b = 0; b < 128; ++b) {
    St s = {{0, 0, 0, 0}};
    s.v[b >> 5] = 1u << (b & 31);
    host_step(s);
    g_pow2[0].col[b] = s;
  }
  for (int k = 1; k < 64; ++k)
    for (int b = 0; b < 128; ++b) g_pow2[k].col[b] = mat_apply(g_pow2[k - 1], g_pow2[k - 1].col[b]);

  // characteristic polynomial via Berlekamp-Massey on one output bit
  St s = {{123456789u, 362436069u, 521288629u, 88675123u}};
  std::vector<int> seq(512);
  for (int i = 0; i < 512; ++i) { host_step(s); seq[i] = s.v[3] & 1u; }
  std::vector<int> C;
  int L = berlekamp_massey(seq, C);
  if (L != 128) { err = "xorshift128: unexpected linear complexity"; return false; }
  memset(&g_P, 0, sizeof(g_P));
  for (int i = 0; i <= 128; ++i)
    if (C[i]) poly_flip(g_P, 128 - i);  // P(x) = x^L C(1/x)

  // self-check: polynomial jump == matrix jump
  uint64_t n = 12ull * 37ull * 1001ull + 5;
  St a = {{123456789u, 362436069u, 521288629u, 88675123u}};
  {
    uint64_t m = n;
    for (int k = 0; m; ++k, m >>= 1)
      if (m & 1) a = mat_apply(g_pow2[k], a);
  }
  Poly c = poly_xpow(n);
  St e0 = {{123456789u, 362436069u, 521288629u, 88675123u}};
  St acc = {{0, 0, 0, 0}};
  for (int i = 0; i < 128; ++i) {
    if (poly_bit(c, i))
      for (int w = 0; w < 4; ++w) acc.v[w] ^= e0.v[w];
    host_step(e0);
  }
  if (memcmp(&a, &acc, sizeof(St)) != 0) { err = "xorshift128 jump tables failed self-check"; return false; }
  g_tables_ready = true;
  return true;
}

std::string g_create_err;

}  // namespace

// ================================================================ shared kernels
// offsets[i] = exclusive prefix sum of counts (single block, chunk per thread)
__global__ __launch_bounds__(KWY_THREADS) void k_scan_u32(const uint32_t *__restrict__ counts,
                                                         uint64_t *__restrict__ offsets, int64_t n) {
  __shared__ uint64_t tot[KWY_THREADS];
  const int t = threadIdx.x;
  const int64_t chunk = (n + KWY_THREADS - 1) / KWY_THREADS;
  const int64_t b0 = t * chunk, b1 = min(n, b0 + chunk);
  uint64_t run = 0;
  for (int64_t i = b0; i < b1; ++i) run += counts[i];
  tot[t] = run;
  __syncthreads();
  if (t == 0) {
    uint64_t acc = 0;
    for (int i = 0; i < KWY_THREADS; ++i) { uint64_t v = tot[i]; tot[i] = acc; acc += v; }
  }
  __syncthreads();
  run = tot[t];
  for (int64_t i = b0; i < b1; ++i) { offsets[i] = run; run += counts[i]; }
  if (b0 < n && b1 == n) offsets[n] = run;
  if (n == 0 && t == 0) offsets[0] = 0;
}

// Per item: jump the seed state by 12*(base+offsets[i]) steps, then emit the 131-word extended sequence of that
// state.  The jump is wavefront-cooperative (128 matrix columns over 64 lanes), the emission is a serial recurrence
// that one lane has to run: a workgroup takes KWY_EBASE_ITEMS items, every wavefront jumps a quarter of them one
// after the other and leaves the states in LDS, then KWY_EBASE_ITEMS lanes of wavefront 0 emit all sequences side by
// side (one lane per item used to emit alone while its 63 neighbours idled: 2.5x the instructions).
#define KWY_EBASE_ITEMS 16
__global__ __launch_bounds__(KWY_THREADS) void k_rng_ebase(const uint64_t *__restrict__ offsets,
                                                          const uint64_t *__restrict__ base_ptr, int64_t n,
                                                          const uint4 *__restrict__ pow2,
                                                          uint32_t *__restrict__ ebase) {
  __shared__ uint32_t st[KWY_EBASE_ITEMS][4];
  __shared__ int live[KWY_EBASE_ITEMS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t item0 = (int64_t)blockIdx.x * KWY_EBASE_ITEMS;
  const uint64_t base_draws = base_ptr ? *base_ptr : 0ull;
  constexpr int PER_WAVE = KWY_EBASE_ITEMS / KWY_WAVES;
  for (int f = 0; f < PER_WAVE; ++f) {
    const int slot = wv * PER_WAVE + f;
    const int64_t item = item0 + slot;
    const bool has = item < n && offsets[item] != ~0ull;   // ~0: item without draws
    if (has) {
      uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
      kwy_wave_jump(s, 12ull * (base_draws + offsets[item]), pow2);
      if (lane == 0) { st[slot][0] = s[0]; st[slot][1] = s[1]; st[slot][2] = s[2]; st[slot][3] = s[3]; }
    }
    if (lane == 0) live[slot] = has ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x < KWY_EBASE_ITEMS && live[threadIdx.x]) {
    kwy_rng r = {st[threadIdx.x][0], st[threadIdx.x][1], st[threadIdx.x][2], st[threadIdx.x][3]};
    uint32_t *e = ebase + (item0 + threadIdx.x) * KWY_EBASE_WORDS;
    e[0] = r.x; e[1] = r.y; e[2] = r.z; e[3] = r.w;
    for (int i = 4; i < 131; ++i) e[i] = kwy_rng_step(r);
    e[131] = 0;
  }
}

int kwy_launch_scan(kwy_ctx *ctx, const uint32_t *counts, uint64_t *offsets, int64_t n) {
  hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, counts, offsets, n);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

int kwy_launch_ebase(kwy_ctx *ctx, const uint64_t *offsets, const uint64_t *base_draws, int64_t n,
                     uint32_t *ebase) {
  if (n <= 0) return KWY_OK;
  int blocks = (int)((n + KWY_EBASE_ITEMS - 1) / KWY_EBASE_ITEMS);
  hipLaunchKernelGGL(k_rng_ebase, dim3(blocks), dim3(KWY_THREADS), 0, ctx->stream, offsets,
                     base_draws, n, ctx->d_pow2, ebase);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// ================================================================ arena
int kwy_arena_begin(kwy_ctx *ctx, size_t bytes) {
  bytes = kwy_pad(bytes) + 4096;
  if (bytes > ctx->arena_cap) {
    KWY_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->arena) KWY_HIP(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_cap = 0;
    size_t cap = bytes + bytes / 4;
    KWY_HIP(hipMalloc((void **)&ctx->arena, cap));
    ctx->arena_cap = cap;
    ++ctx->arena_generation;
  }
  ctx->arena_off = 0;
  return KWY_OK;
}

void *kwy_arena_alloc(kwy_ctx *ctx, size_t bytes) {
  bytes = kwy_pad(bytes);
  if (ctx->arena_off + bytes > ctx->arena_cap) return nullptr;  // programming error: begin() undersized
  void *p = ctx->arena + ctx->arena_off;
  ctx->arena_off += bytes;
  return p;
}

// ================================================================ tables
int kwy_get_twiddles(kwy_ctx *ctx, int log2n, const kwy_c **out) {
  if (log2n < 0 || log2n >= 20) { ctx->err = "fft size out of range"; return KWY_EINVAL; }
  if (!ctx->d_tw[log2n]) {
    int n = 1 << log2n;
    std::vector<kwy_c> h(n);
    for (int k = 0; k < n; ++k) {
      double a = -2.0 * M_PI * (double)k / (double)n;
      h[k].x = cos(a);
      h[k].y = sin(a);
    }
    kwy_c *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(kwy_c) * n));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(kwy_c) * n, hipMemcpyHostToDevice));
    ctx->d_tw[log2n] = d;
  }
  *out = ctx->d_tw[log2n];
  return KWY_OK;
}

int kwy_get_poly(kwy_ctx *ctx, uint64_t stride_steps, const uint4 **out) {
  auto it = ctx->d_poly.find(stride_steps);
  if (it == ctx->d_poly.end()) {
    std::vector<uint4> h(KWY_THREADS);
    Poly step = poly_xpow(stride_steps);
    Poly cur = {{1, 0, 0, 0}};
    for (int t = 0; t < KWY_THREADS; ++t) {
      h[t].x = (uint32_t)cur.w[0]; h[t].y = (uint32_t)(cur.w[0] >> 32);
      h[t].z = (uint32_t)cur.w[1]; h[t].w = (uint32_t)(cur.w[1] >> 32);
      cur = poly_mulmod(cur, step);
    }
    uint4 *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(uint4) * KWY_THREADS));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(uint4) * KWY_THREADS, hipMemcpyHostToDevice));
    it = ctx->d_poly.emplace(stride_steps, d).first;
  }
  *out = it->second;
  return KWY_OK;
}

int kwy_get_poly_multi(kwy_ctx *ctx, int max_c, int nthreads, const uint4 **out) {
  const uint64_t key = 0x8000000000000000ull | ((uint64_t)nthreads << 32) | (uint64_t)max_c;
  auto it = ctx->d_poly.find(key);
  if (it == ctx->d_poly.end()) {
    std::vector<uint4> h((size_t)max_c * nthreads);
    for (int c = 1; c <= max_c; ++c) {
      Poly step = poly_xpow(12ull * c);
      Poly cur = {{1, 0, 0, 0}};
      for (int t = 0; t < nthreads; ++t) {
        uint4 &o = h[(size_t)(c - 1) * nthreads + t];
        o.x = (uint32_t)cur.w[0]; o.y = (uint32_t)(cur.w[0] >> 32);
        o.z = (uint32_t)cur.w[1]; o.w = (uint32_t)(cur.w[1] >> 32);
        cur = poly_mulmod(cur, step);
      }
    }
    uint4 *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(uint4) * h.size()));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(uint4) * h.size(), hipMemcpyHostToDevice));
    it = ctx->d_poly.emplace(key, d).first;
  }
  *out = it->second;
  return KWY_OK;
}

// ================================================================ C ABI: context
extern "C" {

int kwy_version(void) { return 1; }

const char *kwy_create_error(void) { return g_create_err.c_str(); }

int kwy_ctx_create(int device, void *stream, kwy_ctx **out) {
  if (!out) return KWY_EINVAL;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = "libkwy: no HIP device available (the HIP path has no CPU fallback)";
    return KWY_ENODEV;
  }
  if (device < 0 || device >= ndev) { g_create_err = "libkwy: bad device ordinal"; return KWY_EINVAL; }
  if (!build_host_tables(g_create_err)) return KWY_EINVAL;
  kwy_ctx *ctx = new kwy_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; delete ctx; return KWY_EHIP; }
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      g_create_err = "hipStreamCreate failed"; delete ctx; return KWY_EHIP;
    }
    ctx->own_stream = true;
  }
  // jump matrices
  std::vector<uint4> h(64 * 128);
  for (int k = 0; k < 64; ++k)
    for (int b = 0; b < 128; ++b) {
      const St &c = g_pow2[k].col[b];
      h[k * 128 + b] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
    }
  if (hipMalloc((void **)&ctx->d_pow2, sizeof(uint4) * h.size()) != hipSuccess ||
      hipMemcpy(ctx->d_pow2, h.data(), sizeof(uint4) * h.size(), hipMemcpyHostToDevice) != hipSuccess) {
    g_create_err = "device table upload failed";
    delete ctx;
    return KWY_EHIP;
  }
  *out = ctx;
  return KWY_OK;
}

void kwy_ctx_destroy(kwy_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->d_pow2) (void)hipFree(ctx->d_pow2);
  for (auto &p : ctx->d_tw)
    if (p) (void)hipFree(p);
  for (auto &kv : ctx->d_poly) (void)hipFree(kv.second);
  for (auto &kv : ctx->d_mats) (void)hipFree(kv.second);
  for (auto &kv : ctx->prof_events)
    for (auto &ev : kv.second) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int kwy_ctx_sync(kwy_ctx *ctx) {
  if (!ctx) return KWY_EINVAL;
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}

int64_t kwy_ctx_arena_generation(kwy_ctx *ctx) { return ctx ? ctx->arena_generation : -1; }

int kwy_ctx_reserve(kwy_ctx *ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return KWY_EINVAL;
  KWY_HIP(hipSetDevice(ctx->device));
  return kwy_arena_begin(ctx, (size_t)bytes);
}

int kwy_ctx_debug_buffer(kwy_ctx *ctx, void *device_buffer) {
  if (!ctx) return KWY_EINVAL;
  ctx->dbg = device_buffer;
  return KWY_OK;
}

int kwy_ctx_profile(kwy_ctx *ctx, int enable) {
  if (!ctx) return KWY_EINVAL;
  ctx->prof = enable != 0;
  return KWY_OK;
}

int kwy_ctx_profile_read(kwy_ctx *ctx, const char *kernel, double *total_ms, int64_t *count) {
  if (!ctx || !kernel || !total_ms || !count) return KWY_EINVAL;
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  int64_t n = 0;
  auto it = ctx->prof_events.find(kernel);
  if (it != ctx->prof_events.end()) {
    for (auto &ev : it->second) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { tot += ms; ++n; }
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    ctx->prof_events.erase(it);
  }
  *total_ms = tot;
  *count = n;
  return KWY_OK;
}

void *kwy_ctx_stream(kwy_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

const char *kwy_last_error(kwy_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int kwy_cheaptrick_fft_size(int fs, double f0_floor) {
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / f0_floor + 1) / 0.69314718055994529));
}

int64_t kwy_dio_frames(int fs, int64_t x_length, double frame_period_ms) {
  return (int64_t)(1000.0 * x_length / fs / frame_period_ms) + 1;
}

int64_t kwy_synth_length(int64_t f0_length, double frame_period_ms, int fs) {
  return (int64_t)(f0_length * frame_period_ms * fs / 1000);
}

}  // extern "C"
