// kwy_internal.hpp -- host-side internals of libkwy.so (context, scratch arena,
// constant tables).  Not part of the ABI (see include/kwy.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/kwy.h"
#include "kwy_device.hpp"

struct kwy_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // scratch arena (grow-only bump allocator, reset at the start of every call)
  char *arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;
  int64_t arena_generation = 0;            // counts (re)allocations: a captured HIP graph holds arena addresses

  // constant tables
  uint4 *d_pow2 = nullptr;                 // [64][128] columns of T^(2^k)
  const uint32_t *d_randn = nullptr;       // the device's table of WORLD's randn stream (shared by all contexts)
  uint64_t randn_n = 0;                    // draws of it this context uses (kwy_ctx_set_randn_limit lowers it)
  kwy_c *d_tw[20] = {nullptr};             // d_tw[l]: exp(-2 pi i k / 2^l), k < 2^l
  std::map<uint64_t, uint4 *> d_poly;      // stride(steps) -> x^(stride*t) mod P, t < 256
  std::map<std::string, double *> d_mats;  // cached host-built matrices (mcep etc.)
  std::map<std::string, int64_t> i_vals;   // small cached integers that go with them

  // optional per-kernel timing with HIP events on this context's stream
  void *dbg = nullptr;  // optional device buffer for in-kernel cycle stamps (diagnostic builds)
  bool prof = false;
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_events;
};

#define KWY_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return e_ == hipErrorOutOfMemory ? KWY_ENOMEM : KWY_EHIP;                    \
    }                                                                              \
  } while (0)

#define KWY_TRY(call)              \
  do {                             \
    int rc_ = (call);              \
    if (rc_ != KWY_OK) return rc_; \
  } while (0)

// --- profiling ---------------------------------------------------------------
// KWY_PROF(ctx, "kernel", launch-statement): brackets the launch with two HIP
// events when profiling is on (kwy_ctx_profile); otherwise just launches.
struct kwy_prof_scope {
  kwy_ctx *ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char *name;
  kwy_prof_scope(kwy_ctx *c, const char *n) : ctx(c), name(n) {
    if (ctx->prof) {
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
      (void)hipEventRecord(a, ctx->stream);
    }
  }
  ~kwy_prof_scope() {
    if (a && b) {
      (void)hipEventRecord(b, ctx->stream);
      ctx->prof_events[name].push_back({a, b});
    }
  }
};
#define KWY_PROF(ctx, name, stmt) do { kwy_prof_scope ps_(ctx, name); stmt; } while (0)

// --- arena ------------------------------------------------------------------
// Reserve `bytes` of scratch for the current call (must be called once, before
// any kwy_arena_alloc of the call); grows the arena if needed.
int kwy_arena_begin(kwy_ctx *ctx, size_t bytes);
void *kwy_arena_alloc(kwy_ctx *ctx, size_t bytes);
template <typename T>
static inline T *kwy_arena(kwy_ctx *ctx, size_t n) {
  return (T *)kwy_arena_alloc(ctx, n * sizeof(T));
}
static inline size_t kwy_pad(size_t b) { return (b + 255) & ~(size_t)255; }

// --- tables -----------------------------------------------------------------
int kwy_get_twiddles(kwy_ctx *ctx, int log2n, const kwy_c **out);
int kwy_get_poly(kwy_ctx *ctx, uint64_t stride_steps, const uint4 **out);
// pysptk.sp2mc's frequency transform for transform length N as a matrix [ncut][64] (kwy_mcep.hip): mc[j] = sum_n F[n][j] c[n]
int kwy_get_sp2mc_matrix(kwy_ctx *ctx, int N, int order, double alpha, const double **out, int *ncut);
// kwy_gmm_em_sums_dev whose kernels return at once when gate[0] != 0 (gate may be NULL); with `labels` the weights are
// the one-hot rows of the labels (resp is not read): kwy_gmmfit.hip
int kwy_fit_sums_gated(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, const double *resp, double *stats,
                       const long long *gate, const int *labels);
// [max_c][nthreads] table: row c-1 holds x^(12*c*t) mod P, t < nthreads (chunk of c draws per thread)
int kwy_get_poly_multi(kwy_ctx *ctx, int max_c, int nthreads, const uint4 **out);

// the context's view of the randn stream: table, usable length, jump matrices
static inline kwy_randn_src kwy_randn(const kwy_ctx *ctx) {
  return kwy_randn_src{ctx->d_randn, ctx->randn_n, ctx->d_pow2};
}

// --- utterance batches -------------------------------------------------------------
// The frame-parallel analysis kernels take up to KWY_BATCH_MAX utterances per launch: workgroup g belongs to the
// utterance u with start[u] <= g < start[u + 1] and handles its frame g - start[u].  The per-utterance pointers travel
// BY VALUE in the kernel arguments (scalar loads, no descriptor table in device memory, capturable in HIP graphs);
// one launch over both utterances of a pair -- or over all pairs of a step -- fills the chip where a single
// utterance's 2 000 frames are 2.2 rounds of resident workgroups.
#define KWY_BATCH_MAX 16
template <class VIEW>
struct kwy_batch {
  int n;
  int start[KWY_BATCH_MAX + 1];
  VIEW u[KWY_BATCH_MAX];
  // utterance of workgroup g (uniform: scalar loop)
  __device__ __forceinline__ int find(int g) const {
    int k = 0;
    while (k + 1 < n && g >= start[k + 1]) ++k;
    return k;
  }
};

static inline int kwy_ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}
