// kwy_mlsa.hip -- MLSA differential-spectrum filter on gfx950.
//
// Replaces the pysptk calls of kwiiyatta/filter/mlsa.py:24-29:
//     b = pysptk.mc2b(mc, alpha)
//     Synthesizer(MLSADF(order, alpha), hopsize).synthesis(wav, b)
// (SPTK mlsadf: Pade approximant of order pd of exp F(z); per-sample linear interpolation of
// the filter coefficients inside a frame; input scaled by exp(b[0]).)
//
// The filter is a time-varying IIR: serial in the sample index.  One wavefront filters one
// signal; what parallelism there is inside a sample is spread over lanes:
//   * lane k (k <= order) owns coefficient k: interpolation b[k] += slope[k]
//   * lane s (s < pd) owns Pade section s+1 of mlsadf2: its all-pass chain state d[0..m+1]
//     (in LDS, two copies alternating per sample, which replaces SPTK's shift loop) and runs
//     mlsafir() on it -- the pd sections of one sample are independent, each consumes the
//     previous sample's output of the section before it (DPP shift)
//   * the short mlsadf1 cascade and the Pade sums are uniform and computed by every lane
// The waveform is staged through LDS one frame at a time.  Operation order per sample follows
// SPTK exactly; only exp() differs from the CPU's libm in the last bit.
#include <math.h>

#include "kwy_internal.hpp"

#define MLSA_MAX_ORDER 62
#define MLSA_MAX_HOP 2048

__constant__ double c_pade[21] = {1.0,
                                  1.0, 0.0,
                                  1.0, 0.0, 0.0,
                                  1.0, 0.0, 0.0, 0.0,
                                  1.0, 0.4999273, 0.1067005, 0.01170221, 0.0005656279,
                                  1.0, 0.4999391, 0.1107098, 0.01369984, 0.0009564853, 0.00003041721};

// One signal of a launch: the jobs travel by value in the kernel arguments (capturable, no descriptor in memory).
#define MLSA_JOBS_MAX 64
struct mlsa_job {
  const double *x;    // n samples in
  const double *mc;   // T x (m + 1): mel-cepstra (k_mc2b) -- or the filter coefficients b themselves (k_mlsa_filter)
  double *b;          // T x (m + 1): filter coefficients (k_mc2b's output)
  double *y;          // n samples out
  int64_t n, T;
};
struct mlsa_jobs {
  int count;
  mlsa_job j[MLSA_JOBS_MAX];
};

// b[m] = c[m]; b[i] = c[i] - a b[i+1]   (pysptk.mc2b, one thread per frame); zero_c0: c[0] is taken as 0
// (apply_mlsa_filter's `mcep.data[:, 0] = 0`: the filter changes the envelope's shape, not its power)
__global__ void k_mc2b(mlsa_jobs J, int m, double a, int zero_c0) {
  const mlsa_job q = J.j[blockIdx.y];
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= q.T) return;
  const double *c = q.mc + t * (m + 1);
  double *o = q.b + t * (m + 1);
  double nxt = c[m];
  o[m] = nxt;
  for (int i = m - 1; i >= 0; --i) { nxt = ((i == 0 && zero_c0) ? 0.0 : c[i]) - a * nxt; o[i] = nxt; }
}

// NB = number of 8-link blocks of the all-pass chain (links i = 2 .. 8 NB + 1 >= m; the coefficients
// of the links behind m are zero).  The chain state lives in registers: the loops over links are
// unrolled, SPTK's shift d[i] = d[i-1] becomes the register the new value is written to.
// One wavefront (workgroup) per signal of the launch.
template <int NB, int PD>
__global__ __launch_bounds__(64) void k_mlsa_filter(mlsa_jobs J, int m, double a, int hop) {
  const double *__restrict__ x = J.j[blockIdx.x].x;
  const double *__restrict__ b = J.j[blockIdx.x].b;
  double *__restrict__ y = J.j[blockIdx.x].y;
  const int64_t n = J.j[blockIdx.x].n, T = J.j[blockIdx.x].T;
  constexpr int pd = PD;
  constexpr int ML = 8 * NB;
  extern __shared__ double sm[];
  // sb[2][72] interpolated coefficients of this and the next sample, sx[hop] in, sy[hop] out
  double *sb = sm;
  double *sx = sb + 2 * 72;
  double *sy = sx + hop;
  const int lane = threadIdx.x;
  const double aa = 1 - a * a;
  const double *ppade = &c_pade[pd * (pd + 1) / 2];
  for (int i = lane; i < 2 * 72; i += 64) sb[i] = 0.0;
  // frame f is processed iff (f+1) hop < n; what lies behind the last processed frame stays zero
  int64_t nproc = (n - 1) / hop;
  if (nproc > T) nproc = T;
  for (int64_t j = nproc * hop + lane; j < n; j += 64) y[j] = 0.0;
  // mlsadf1 state (uniform): d1[1..pd], pt1[0..pd]
  double d1[PD + 1], pt1[PD + 1];
#pragma unroll
  for (int i = 0; i <= PD; ++i) d1[i] = pt1[i] = 0.0;
  // mlsadf2: lane s < pd owns Pade section s+1 and its chain d[0..m+1]
  double d[ML + 3];
#pragma unroll
  for (int i = 0; i < ML + 3; ++i) d[i] = 0.0;
  double pt0 = 0.0;   // pt[0] of mlsadf2: the section input of the previous sample
  double po = 0.0;    // this lane's section output of the previous sample
  double prevb = (lane <= m) ? b[lane] : 0.0;  // coefficient `lane` of the previous frame (frame 0: its own)
  int par = 0;
  __syncthreads();
  for (int64_t f = 0; f < nproc; ++f) {
    const int64_t s0 = f * hop;
    const double curb = (lane <= m) ? b[f * (m + 1) + lane] : 0.0;
    const double slope = (curb - prevb) / hop;
    double cur = prevb;
    // The input scale exp(b[0]) of every sample of the frame, off the serial loop: b[0] at sample j is prevb[0] plus j
    // additions of the slope (the very sums the loop below forms for its coefficients), so every lane runs those
    // additions and keeps the values of "its" samples j = lane + 64 r -- then exp and the product in parallel, the
    // same numbers the loop used to compute one sample at a time (~40 instructions of each sample's ~280).
    {
      const double p0 = kwy_readlane_f64(prevb, 0), sl0 = kwy_readlane_f64(slope, 0);
      double v = p0;
      for (int t = 0; t < lane; ++t) v += sl0;                  // sample j = lane
      for (int j = lane; j < hop; j += 64) {
        sx[j] = x[s0 + j] * exp(v);
        for (int t = 0; t < 64; ++t) v += sl0;                  // 64 samples on
      }
    }
    if (lane <= m) sb[par * 72 + lane] = cur;
    __syncthreads();
    for (int j = 0; j < hop; ++j) {
      // one wavefront: the LDS executes its accesses in issue order, no barrier between the
      // store of a coefficient and its broadcast load
      const double *__restrict__ bc = sb + par * 72;
      const double b1 = bc[1];
      cur += slope;
      if (lane <= m) sb[(par ^ 1) * 72 + lane] = cur;   // the coefficients of the next sample
      double xv = sx[j];                                 // (already times exp(b[0]) of this sample)
      // ---- mlsadf1 (uniform)
      double out = 0.0;
#pragma unroll
      for (int i = PD; i >= 1; --i) {
        d1[i] = aa * pt1[i - 1] + a * d1[i];
        pt1[i] = d1[i] * b1;
        const double v = pt1[i] * ppade[i];
        xv += (1 & i) ? v : -v;
        out += v;
      }
      pt1[0] = xv;
      out += xv;
      const double x2 = out;
      // ---- mlsadf2: section lane+1 runs mlsafir on its chain
      double in = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(po), 0x138, 0xf, 0xf, false),
                                   __builtin_amdgcn_update_dpp(0, __double2loint(po), 0x138, 0xf, 0xf, false));
      if (lane == 0) in = pt0;
      if (lane >= pd) in = 0.0;   // idle lanes keep an all-zero chain
      // SPTK: d[0] = x; d[1] = aa d[0] + a d[1]; d[i] += a (d[i+1] - d[i-1]) (i = 2..m), y += d[i] b[i];
      //       then d[i] = d[i-1] (i = m+1..2): the new d[i] is written to d[i+1] straight away.
      double yo = 0.0;
      double u_prev = aa * in + a * d[1];  // new d[1]
      double di = d[2];
      d[0] = in;
      d[1] = u_prev;
      d[2] = u_prev;                        // the shift copies d[1] into d[2] as well
      // coefficients of block k+1 are fetched (LDS broadcast loads) while the links of block k run
      double bb[8], bn[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) bb[q] = bc[2 + q];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (k + 1 < NB) {
#pragma unroll
          for (int q = 0; q < 8; ++q) bn[q] = bc[2 + 8 * (k + 1) + q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = 2 + 8 * k + q;
          const double dn = d[i + 1];
          const double u = di + a * (dn - u_prev);
          yo += u * bb[q];
          d[i + 1] = u;
          u_prev = u;
          di = dn;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) bb[q] = bn[q];
      }
      // ---- Pade sums (uniform): sections pd .. 1
      double xx = x2, out2 = 0.0;
#pragma unroll
      for (int i = PD; i >= 1; --i) {
        const double pti = kwy_readlane_f64(yo, i - 1);
        const double v = pti * ppade[i];
        xx += (1 & i) ? v : -v;
        out2 += v;
      }
      pt0 = xx;
      out2 += xx;
      po = yo;
      if (lane == 0) sy[j] = out2;
      par ^= 1;
      __builtin_amdgcn_wave_barrier();
    }
    for (int j = lane; j < hop; j += 64) y[s0 + j] = sy[j];
    prevb = curb;
    __syncthreads();
  }
}

static int mlsa_check(kwy_ctx *ctx, const void *x, int64_t n, const void *b, int64_t T, int m, double a, int pd,
                      int hop, const void *y) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !b || !y || n <= 0 || T <= 0 || m < 2 || m > MLSA_MAX_ORDER || !(fabs(a) < 1.0) || pd < 4 || pd > 5 ||
      hop < 1 || hop > MLSA_MAX_HOP) {
    ctx->err = "mlsa_synthesis: bad argument (order 2..62, pd 4 or 5, hopsize 1..2048)";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

template <int NB>
static int mlsa_launch(kwy_ctx *ctx, const mlsa_jobs &J, int m, double a, int pd, int hop) {
  const size_t lds = sizeof(double) * (2 * 72 + 2 * hop);
  void (*kern)(mlsa_jobs, int, double, int) = (pd == 4) ? k_mlsa_filter<NB, 4> : k_mlsa_filter<NB, 5>;
  KWY_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_mlsa_filter", hipLaunchKernelGGL(kern, dim3(J.count), dim3(64), lds, ctx->stream, J, m, a, hop));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// the filters of J.count signals (J.j[].b: their coefficients), one workgroup each
static int mlsa_core(kwy_ctx *ctx, const mlsa_jobs &J, int m, double a, int pd, int hop) {
  switch ((m - 1 + 7) / 8) {   // links 2..m in blocks of 8
    case 1: return mlsa_launch<1>(ctx, J, m, a, pd, hop);
    case 2: return mlsa_launch<2>(ctx, J, m, a, pd, hop);
    case 3: return mlsa_launch<3>(ctx, J, m, a, pd, hop);
    case 4: return mlsa_launch<4>(ctx, J, m, a, pd, hop);
    case 5: return mlsa_launch<5>(ctx, J, m, a, pd, hop);
    case 6: return mlsa_launch<6>(ctx, J, m, a, pd, hop);
    case 7: return mlsa_launch<7>(ctx, J, m, a, pd, hop);
    default: return mlsa_launch<8>(ctx, J, m, a, pd, hop);
  }
}

static mlsa_jobs mlsa_one(const double *x, int64_t n, const double *mc, double *b, int64_t T, double *y) {
  mlsa_jobs J;
  J.count = 1;
  J.j[0] = mlsa_job{x, mc, b, y, n, T};
  return J;
}

static int mc2b_launch(kwy_ctx *ctx, const mlsa_jobs &J, int64_t T_max, int order, double alpha, int zero_c0) {
  hipLaunchKernelGGL(k_mc2b, dim3((unsigned)((T_max + 255) / 256), J.count), dim3(256), 0, ctx->stream, J, order, alpha,
                     zero_c0);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_mc2b_dev(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, double *b) {
  if (!ctx) return KWY_EINVAL;
  if (!mc || !b || T <= 0 || order < 1) { ctx->err = "mc2b: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  return mc2b_launch(ctx, mlsa_one(nullptr, 0, mc, b, T, nullptr), T, order, alpha, 0);
}

extern "C" int kwy_mc2b(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, double *b) {
  if (!ctx) return KWY_EINVAL;
  if (!mc || !b || T <= 0 || order < 1) { ctx->err = "mc2b: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const size_t bm = kwy_pad(sizeof(double) * T * (order + 1));
  KWY_TRY(kwy_arena_begin(ctx, 2 * bm));
  double *dmc = kwy_arena<double>(ctx, (size_t)T * (order + 1)), *db = kwy_arena<double>(ctx, (size_t)T * (order + 1));
  KWY_HIP(hipMemcpyAsync(dmc, mc, sizeof(double) * T * (order + 1), hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_mc2b_dev(ctx, dmc, T, order, alpha, db));
  KWY_HIP(hipMemcpyAsync(b, db, sizeof(double) * T * (order + 1), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}

extern "C" int kwy_mlsa_synthesis_dev(kwy_ctx *ctx, const double *x, int64_t n, const double *b, int64_t T,
                                      int order, double alpha, int pd, int hopsize, double *y) {
  KWY_TRY(mlsa_check(ctx, x, n, b, T, order, alpha, pd, hopsize, y));
  KWY_HIP(hipSetDevice(ctx->device));
  return mlsa_core(ctx, mlsa_one(x, n, nullptr, (double *)b, T, y), order, alpha, pd, hopsize);
}

// apply_mlsa_filter for a batch of signals (device pointers, not synchronised): mc2b of every job's mel-cepstra
// (ignore_c0: with c0 taken as zero) into the context's scratch, then the filters, one wavefront per signal,
// <= MLSA_JOBS_MAX signals per launch.  Every job equals kwy_mc2b_dev + kwy_mlsa_synthesis_dev bit for bit.
extern "C" int kwy_mlsa_filter_batch_dev(kwy_ctx *ctx, const kwy_mlsa_job *jobs, int count, int order, double alpha,
                                         int pd, int hopsize, int ignore_c0) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 1) { ctx->err = "mlsa_filter_batch: bad argument"; return KWY_EINVAL; }
  size_t need = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_mlsa_job &q = jobs[i];
    KWY_TRY(mlsa_check(ctx, q.x, q.x_length, q.mc, q.T, order, alpha, pd, hopsize, q.y));
    need += kwy_pad(sizeof(double) * (size_t)q.T * (order + 1));
  }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, need));
  for (int i0 = 0; i0 < count; i0 += MLSA_JOBS_MAX) {
    mlsa_jobs J;
    J.count = count - i0 < MLSA_JOBS_MAX ? count - i0 : MLSA_JOBS_MAX;
    int64_t T_max = 0;
    for (int u = 0; u < J.count; ++u) {
      const kwy_mlsa_job &q = jobs[i0 + u];
      double *b = kwy_arena<double>(ctx, (size_t)q.T * (order + 1));
      if (!b) { ctx->err = "mlsa_filter_batch: scratch arena too small"; return KWY_ENOMEM; }
      J.j[u] = mlsa_job{q.x, q.mc, b, q.y, q.x_length, q.T};
      T_max = q.T > T_max ? q.T : T_max;
    }
    KWY_TRY(mc2b_launch(ctx, J, T_max, order, alpha, ignore_c0 ? 1 : 0));
    KWY_TRY(mlsa_core(ctx, J, order, alpha, pd, hopsize));
  }
  return KWY_OK;
}

extern "C" int kwy_mlsa_synthesis(kwy_ctx *ctx, const double *x, int64_t n, const double *b, int64_t T, int order,
                                  double alpha, int pd, int hopsize, double *y) {
  KWY_TRY(mlsa_check(ctx, x, n, b, T, order, alpha, pd, hopsize, y));
  KWY_HIP(hipSetDevice(ctx->device));
  const size_t bx = kwy_pad(sizeof(double) * n), bm = kwy_pad(sizeof(double) * T * (order + 1));
  KWY_TRY(kwy_arena_begin(ctx, 2 * bx + bm));
  double *dx = kwy_arena<double>(ctx, n), *dy = kwy_arena<double>(ctx, n);
  double *db = kwy_arena<double>(ctx, (size_t)T * (order + 1));
  KWY_HIP(hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(db, b, sizeof(double) * T * (order + 1), hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(mlsa_core(ctx, mlsa_one(dx, n, nullptr, db, T, dy), order, alpha, pd, hopsize));
  KWY_HIP(hipMemcpyAsync(y, dy, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
