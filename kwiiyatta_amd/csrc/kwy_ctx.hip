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
#include <stdlib.h>
#include <string.h>

#include <mutex>

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

// ================================================================ the randn table
// The first 2^L raw draws of WORLD's stream, once per device and process (kwy_device.hpp: kwy_randn_src).  A
// workgroup fills KWY_TABLE_CHUNK consecutive draws: it jumps to the chunk's stream position, every thread derives the
// state of "its" 16 draws from the chunk's extended sequence (jump table in LDS) and steps through them.
#define KWY_TABLE_PER_THREAD 16
#define KWY_TABLE_CHUNK (KWY_THREADS * KWY_TABLE_PER_THREAD)
__global__ __launch_bounds__(KWY_THREADS) void k_rng_table_fill(uint32_t *__restrict__ tab,
                                                               const uint4 *__restrict__ pow2,
                                                               const uint4 *__restrict__ poly) {
  __shared__ uint32_t e[KWY_EBASE_WORDS];
  __shared__ uint4 jtab[512];
  const uint64_t first = (uint64_t)blockIdx.x * KWY_TABLE_CHUNK;
  kwy_rng_block_ebase(first, pow2, e);
  kwy_rng_build_table<KWY_THREADS>(e, jtab);
  __syncthreads();
  kwy_rng rng = kwy_rng_combine_table(jtab, poly[threadIdx.x]);   // poly: x^(12 * 16 * t) mod P
  uint32_t v[KWY_TABLE_PER_THREAD];
#pragma unroll
  for (int j = 0; j < KWY_TABLE_PER_THREAD; ++j) v[j] = kwy_rng_randn_raw(rng);
  uint4 *o = (uint4 *)(tab + first + (uint64_t)threadIdx.x * KWY_TABLE_PER_THREAD);
#pragma unroll
  for (int q = 0; q < KWY_TABLE_PER_THREAD / 4; ++q) o[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

namespace {
struct RandnTable { uint32_t *d = nullptr; uint64_t n = 0; bool tried = false; };
std::mutex g_randn_mutex;
std::map<int, RandnTable> g_randn;   // per device; lives as long as the process

// log2 of the table length in draws: KWY_RANDN_LOG2 in the environment (12 ... 30; 0 = no table), default 25 =
// 128 MB, which covers D4C of a 10 s utterance at 48 kHz down to a mean f0 of about 70 Hz (8 - 9 M draws at 140 Hz).
int randn_table_log2() {
  const char *v = getenv("KWY_RANDN_LOG2");
  if (!v || !*v) return 25;
  const int l = atoi(v);
  if (l <= 0) return 0;
  return l < 12 ? 12 : (l > 30 ? 30 : l);
}
}  // namespace

static int kwy_attach_randn_table(kwy_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_randn_mutex);
  RandnTable &t = g_randn[ctx->device];
  if (!t.tried) {
    t.tried = true;
    const int l = randn_table_log2();
    if (l > 0) {
      const uint64_t n = 1ull << l;
      const uint4 *poly;
      KWY_TRY(kwy_get_poly(ctx, 12ull * KWY_TABLE_PER_THREAD, &poly));
      uint32_t *d = nullptr;
      if (hipMalloc((void **)&d, sizeof(uint32_t) * n) != hipSuccess) {
        (void)hipGetLastError();   // no room for the table: every draw comes from the jump-ahead path
        ctx->d_randn = nullptr;
        ctx->randn_n = 0;
        return KWY_OK;
      }
      hipLaunchKernelGGL(k_rng_table_fill, dim3((unsigned)(n / KWY_TABLE_CHUNK)), dim3(KWY_THREADS), 0, ctx->stream, d,
                         ctx->d_pow2, poly);
      KWY_HIP(hipGetLastError());
      KWY_HIP(hipStreamSynchronize(ctx->stream));
      t.d = d;
      t.n = n;
    }
  }
  ctx->d_randn = t.d;
  ctx->randn_n = t.n;
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

// ================================================================ device-to-device copy kernel
__global__ __launch_bounds__(KWY_THREADS) void k_copy(double2 *__restrict__ dst, const double2 *__restrict__ src, int64_t n2,
                                                     double *__restrict__ dst1, const double *__restrict__ src1) {
  const int64_t stride = (int64_t)gridDim.x * KWY_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
  if (dst1 && blockIdx.x == 0 && threadIdx.x == 0) *dst1 = *src1;        // the odd eighth byte group
}

// buffers that are only 8-byte aligned (a staging offset behind an odd number of samples): one double per lane
__global__ __launch_bounds__(KWY_THREADS) void k_copy8(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KWY_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// ================================================================ C ABI: context
extern "C" {

int kwy_copy_dev(kwy_ctx *ctx, void *dst, const void *src, int64_t bytes) {
  if (!ctx) return KWY_EINVAL;
  if (!dst || !src || bytes < 0 || (bytes & 7) || ((uintptr_t)dst & 7) || ((uintptr_t)src & 7)) {
    ctx->err = "copy: bad argument (8-byte granularity)";
    return KWY_EINVAL;
  }
  if (bytes == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  if (((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) {      // 8-byte aligned only: a kernel too, one double per lane
    const int64_t n = bytes / 8;
    int64_t g8 = (n + KWY_THREADS * 8 - 1) / (KWY_THREADS * 8);
    g8 = g8 < 1 ? 1 : (g8 > 4096 ? 4096 : g8);
    hipLaunchKernelGGL(k_copy8, dim3((unsigned)g8), dim3(KWY_THREADS), 0, ctx->stream, (double *)dst, (const double *)src, n);
    KWY_HIP(hipGetLastError());
    return KWY_OK;
  }
  const int64_t n2 = bytes / 16;
  const bool odd = (bytes & 15) != 0;
  int64_t g = (n2 + KWY_THREADS * 4 - 1) / (KWY_THREADS * 4);
  g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
  hipLaunchKernelGGL(k_copy, dim3((unsigned)g), dim3(KWY_THREADS), 0, ctx->stream, (double2 *)dst, (const double2 *)src, n2,
                     odd ? (double *)dst + 2 * n2 : nullptr, odd ? (const double *)src + 2 * n2 : nullptr);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

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
  if (kwy_attach_randn_table(ctx) != KWY_OK) {
    g_create_err = "randn table: " + ctx->err;
    kwy_ctx_destroy(ctx);
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

int64_t kwy_ctx_set_randn_limit(kwy_ctx *ctx, int64_t draws) {
  if (!ctx) return -1;
  std::lock_guard<std::mutex> lock(g_randn_mutex);
  const uint64_t have = g_randn[ctx->device].n;
  ctx->randn_n = draws < 0 ? have : ((uint64_t)draws < have ? (uint64_t)draws : have);
  return (int64_t)ctx->randn_n;
}

int kwy_randn_stream(kwy_ctx *ctx, int64_t first, int64_t count, double *out) {
  if (!ctx || !out || first < 0 || count < 0) return KWY_EINVAL;
  if ((uint64_t)(first + count) > ctx->randn_n) { ctx->err = "randn_stream: beyond the table"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  std::vector<uint32_t> raw((size_t)count);
  KWY_HIP(hipMemcpy(raw.data(), ctx->d_randn + first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < count; ++i) out[i] = raw[(size_t)i] / 268435456.0 - 6.0;
  return KWY_OK;
}

// Not part of the ABI (include/kwy.h): tools/d4c_stamps.py and tools/dtw_stamps.py hand instrumented kernels a device
// buffer of >= 64 int64 for clock64() stamps of the workgroup whose index is stored in element 63 (NULL = off).
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
