// kwy_device.hpp -- device-side building blocks shared by the gfx950 kernels:
//   * WORLD's xorshift128 "randn" stream with GF(2) jump-ahead, so that every
//     frame / pulse draws exactly the numbers the serial CPU algorithm would
//   * block-wide f64 reductions and scans (64-wide wavefronts)
//   * LDS-resident double-precision Stockham FFTs (radix-4 + one radix-2 pass)
//     and the half-length real-FFT packing around them
// Written for CDNA4 only (wave64, 160 KB LDS); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KWY_THREADS 256
#define KWY_WAVES (KWY_THREADS / 64)
#define KWY_PI 3.1415926535897932384

// ---------------------------------------------------------------- xorshift128
struct kwy_rng {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ void kwy_rng_seed(kwy_rng &r) {
  r.x = 123456789u; r.y = 362436069u; r.z = 521288629u; r.w = 88675123u;
}

__device__ __forceinline__ uint32_t kwy_rng_step(kwy_rng &r) {
  uint32_t t = r.x ^ (r.x << 11);
  r.x = r.y; r.y = r.z; r.z = r.w;
  r.w = (r.w ^ (r.w >> 19)) ^ (t ^ (t >> 8));
  return r.w;
}

// one "randn" = 12 generator steps (sum of 12 uniforms, WORLD matlabfunctions);
// the raw integer sum keeps a draw in one register, randn = raw / 2^28 - 6
__device__ __forceinline__ uint32_t kwy_rng_randn_raw(kwy_rng &r) {
  uint32_t tmp = kwy_rng_step(r) >> 4;
#pragma unroll
  for (int i = 0; i < 11; ++i) tmp += kwy_rng_step(r) >> 4;
  return tmp;
}
__device__ __forceinline__ double kwy_rng_randn(kwy_rng &r) {
  return kwy_rng_randn_raw(r) / 268435456.0 - 6.0;
}

// XOR over the wavefront, result in every lane: row-local DPP shifts, then the four row totals
// through scalar registers (no LDS round trips).
__device__ __forceinline__ uint32_t kwy_wave_xor_u32(uint32_t v) {
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // lane 15 of a row: the row's XOR
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 15) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 31) ^
         (uint32_t)__builtin_amdgcn_readlane((int)v, 47) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Wavefront-cooperative jump: all 64 lanes hold the same state s[4]; on return
// every lane holds T^steps s.  pow2: [64][128] columns of T^(2^k) (uint4 each).
__device__ inline void kwy_wave_jump(uint32_t (&s)[4], uint64_t steps, const uint4 *__restrict__ pow2) {
  const int lane = threadIdx.x & 63;
  for (int k = 0; steps != 0; ++k, steps >>= 1) {
    if (!(steps & 1)) continue;
    // lane l owns state bits l and l+64 (words l>>5 and 2+(l>>5))
    const uint32_t wlo = lane < 32 ? s[0] : s[1];
    const uint32_t whi = lane < 32 ? s[2] : s[3];
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    uint32_t m = 0u - ((wlo >> (lane & 31)) & 1u);
    uint4 c = pow2[k * 128 + lane];
    o0 ^= c.x & m; o1 ^= c.y & m; o2 ^= c.z & m; o3 ^= c.w & m;
    m = 0u - ((whi >> (lane & 31)) & 1u);
    c = pow2[k * 128 + 64 + lane];
    o0 ^= c.x & m; o1 ^= c.y & m; o2 ^= c.z & m; o3 ^= c.w & m;
    s[0] = kwy_wave_xor_u32(o0); s[1] = kwy_wave_xor_u32(o1);
    s[2] = kwy_wave_xor_u32(o2); s[3] = kwy_wave_xor_u32(o3);
  }
}

// e[0..130]: the extended word sequence of a start state S0 (e0..e3 = x,y,z,w,
// e[4+i] = output of step i+1), so that T^i S0 = (e[i], e[i+1], e[i+2], e[i+3]).
#define KWY_EBASE_WORDS 132
__device__ inline void kwy_rng_ebase(kwy_rng s, uint32_t *e) {
  e[0] = s.x; e[1] = s.y; e[2] = s.z; e[3] = s.w;
  for (int i = 4; i < 131; ++i) e[i] = kwy_rng_step(s);
  e[131] = 0;
}

// state = sum_i c_i T^i S0 with c = coefficients of x^n mod P(x) (P = the
// characteristic polynomial of T); e[] lives in LDS.
__device__ __forceinline__ void kwy_rng_combine_word(kwy_rng &r, const uint32_t *e, uint32_t bits,
                                                      uint32_t &w0, uint32_t &w1, uint32_t &w2) {
#pragma unroll 8
  for (int b = 0; b < 32; ++b) {
    uint32_t w3 = e[b + 3];
    uint32_t m = 0u - ((bits >> b) & 1u);
    r.x ^= w0 & m; r.y ^= w1 & m; r.z ^= w2 & m; r.w ^= w3 & m;
    w0 = w1; w1 = w2; w2 = w3;
  }
}

__device__ __forceinline__ kwy_rng kwy_rng_combine(const uint32_t *e, uint4 c) {
  kwy_rng r = {0, 0, 0, 0};
  uint32_t w0 = e[0], w1 = e[1], w2 = e[2];
  kwy_rng_combine_word(r, e, c.x, w0, w1, w2);
  kwy_rng_combine_word(r, e + 32, c.y, w0, w1, w2);
  kwy_rng_combine_word(r, e + 64, c.z, w0, w1, w2);
  kwy_rng_combine_word(r, e + 96, c.w, w0, w1, w2);
  return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (it is a workgroup-scope fence: s_waitcnt vmcnt(0) before s_barrier) -- a microsecond per barrier in a loop that
// streams results out while it exchanges through LDS.  Use where no thread reads global data another thread of the
// workgroup wrote.
__device__ __forceinline__ void kwy_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- the randn stream as a table
// WORLD reseeds its generator at the entry of CheapTrick, D4C and Synthesis, so draw i of the stream is the same
// number in every call, for every utterance: the first `n` raw 12-step sums live in one process-wide table per
// device (kwy_ctx.hip), and a consumer that knows the stream position of its piece loads the draws -- coalesced
// dwords, shared by all streams through the 256 MiB Infinity Cache -- instead of jumping the generator and stepping
// it (which was a quarter of the D4C stage's instructions).  Pieces that reach beyond the table fall back to the
// GF(2) jump-ahead below, inside the consumer (kwy_rng_block_ebase + jump table + 12 steps per draw): same draws.
struct kwy_randn_src {
  const uint32_t *tab;   // raw sums of draws [0, n)
  uint64_t n;
  const uint4 *pow2;     // [64][128] columns of T^(2^k), for the positions beyond the table
};

__device__ __forceinline__ double kwy_randn_from_raw(uint32_t raw) {
  return __builtin_fma((double)raw, 3.7252902984619140625e-09, -6.0);   // raw / 2^28 - 6: the product is exact, so the
                                                                       // fused form rounds like the division
}

// threadIdx.x behind an optimisation barrier: address arithmetic derived from it is redone where it
// is used instead of being computed once per kernel and carried (spilled) across every phase.
__device__ __forceinline__ int kwy_tid_opaque() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// A value every lane holds identically, moved to scalar registers (so that everything
// derived from it is scalar too and stays out of the vector register budget).
__device__ __forceinline__ double kwy_uniform(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The same combination four coefficient bits at a time: for every nibble position g (32 of
// them) and nibble value v the XOR of the <= 4 selected state vectors is tabulated once per
// stream position (512 entries of 16 B, built by the whole workgroup from e[]), after which a
// thread's jump is 32 table reads + XORs instead of 128 select-and-XOR steps.
template <int NT>
__device__ __forceinline__ void kwy_rng_build_table(const uint32_t *e, uint4 *tab) {
  for (int id = threadIdx.x; id < 512; id += NT) {
    const int g = id >> 4, v = id & 15;
    const uint32_t *q = e + 4 * g;
    uint32_t x = 0, y = 0, z = 0, w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t m = 0u - ((v >> b) & 1u);
      x ^= q[b] & m; y ^= q[b + 1] & m; z ^= q[b + 2] & m; w ^= q[b + 3] & m;
    }
    tab[id] = make_uint4(x, y, z, w);
  }
}

__device__ __forceinline__ kwy_rng kwy_rng_combine_table(const uint4 *tab, uint4 c) {
  kwy_rng r = {0, 0, 0, 0};
  const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const uint4 t = tab[(8 * k + n) * 16 + ((cw[k] >> (4 * n)) & 15u)];
      r.x ^= t.x; r.y ^= t.y; r.z ^= t.z; r.w ^= t.w;
    }
  }
  return r;
}

// Fallback beyond the randn table: the extended word sequence e[0..131] (LDS) of the stream position `pos` (in draws),
// made by the workgroup itself -- wavefront 0 jumps the seed state, its lane 0 runs the 127-step recurrence, the
// others wait.  Slow (about 3 k instruction times), which is why the table should cover the utterance.
// Ends with a barrier.
__device__ inline void kwy_rng_block_ebase(uint64_t pos, const uint4 *__restrict__ pow2, uint32_t *e) {
  if (threadIdx.x < 64) {
    uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
    kwy_wave_jump(s, 12ull * pos, pow2);
    if (threadIdx.x == 0) {
      kwy_rng r = {s[0], s[1], s[2], s[3]};
      kwy_rng_ebase(r, e);
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ cos on [-pi, pi]
// (Its Horner steps as explicit fused multiply-adds were measured in round 3 and dropped: half the instructions of the
// polynomial, but the different schedule pushes k_d4c_body and k_syn_pulse, both at their register cap, from 12 / 96
// to 92 / 224 bytes of scratch per lane: k_d4c_body 0.315 -> 0.334 ms.)
// The analysis windows evaluate cos() a few thousand times per frame with arguments that never
// leave [-pi, pi] (up to rounding).  Two-constant Cody-Waite reduction by pi/2 and the fdlibm
// kernel polynomials: < 1 ulp, a quarter of the instructions of the general-range routine.
__device__ __forceinline__ double kwy_cos_pi_range(double x) {
  const double ax = fabs(x);
  const double kf = rint(ax * 6.36619772367581382433e-01);  // 0, 1 or 2 (3 if x is a hair beyond pi... )
  const int k = (int)kf;
  double y = ax - kf * 1.57079632673412561417e+00;
  y = y - kf * 6.07710050650619224932e-11;
  const double z = y * y;
  // sin kernel
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sn = y + (z * y) * (-1.66666666666666324348e-01 + z * rs);
  // cos kernel
  const double rc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 +
                                   z * (-2.75573143513906633035e-07 +
                                        z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + z * rc);
  // cos(ax) = cos(y + k pi/2)
  const double v = (k & 1) ? sn : cs;
  return ((k + 1) & 2) ? -v : v;
}

// ------------------------------------------------------------------ log, and sincos beyond [-pi, pi]
// The library's log / sincos are 98 / 155 vector instructions a call (extended tables, the Payne-Hanek path for huge
// arguments inline): the kernels that call them once per spectral bin -- CheapTrick's log spectrum, the two half
// log-spectra and the two minimum-phase exponentials of every synthesis pulse -- spend a tenth to a quarter of their
// instructions there.  These are the fdlibm forms (e_log.c; k_sin.c / k_cos.c behind a two-constant fused reduction):
// < 1 ulp for log, < 1.5 ulp for sin / cos with |x| < 2^30 (tests/test_select_gpu.py checks both against numpy).
__device__ __forceinline__ double kwy_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const bool tiny = x < 2.2250738585072014e-308;            // subnormal (or zero / negative: fixed up below)
  const double xs = tiny ? x * 18014398509481984.0 : x;     // 2^54
  int e;
  double m = frexp(xs, &e);                                  // [0.5, 1)
  e -= tiny ? 54 : 0;
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e -= low ? 1 : 0;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  double r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  r = x == 0.0 ? -INFINITY : r;
  r = (x < 0.0 || x != x) ? NAN : r;
  r = x == INFINITY ? INFINITY : r;
  return r;
}

__device__ __forceinline__ void kwy_sincos_medium(double x, double *sn_out, double *cs_out) {
  const double kf = rint(x * 6.36619772367581382433e-01);    // x / (pi/2)
  double y = __builtin_fma(-kf, 1.57079632679489655800e+00, x);
  y = __builtin_fma(-kf, 6.12323399573676603587e-17, y);
  const int q = (int)kf;
  const double z = y * y;
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sn = y + (z * y) * (-1.66666666666666324348e-01 + z * rs);
  const double rc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 +
                                   z * (-2.75573143513906633035e-07 +
                                        z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + z * rc);
  // (sin, cos)(y + q pi/2)
  const double s_ = (q & 1) ? cs : sn, c_ = (q & 1) ? sn : cs;
  *sn_out = (q & 2) ? -s_ : s_;
  *cs_out = ((q + 1) & 2) ? -c_ : c_;
}

// sin and cos on [-pi, pi] from the same reduction (the two kernel polynomials are evaluated by the routine above
// anyway).  A window function sampled at i = tid + NT r advances its argument by a constant per r: one call for
// r = 0, then the rotation kwy_rotate() per further element (6 instead of ~40 instructions; the error grows by about
// an ulp per step, 16 steps at most).
__device__ __forceinline__ void kwy_sincos_pi_range(double x, double *sn_out, double *cs_out) {
  const double ax = fabs(x);
  const double kf = rint(ax * 6.36619772367581382433e-01);
  const int k = (int)kf;
  double y = ax - kf * 1.57079632673412561417e+00;
  y = y - kf * 6.07710050650619224932e-11;
  const double z = y * y;
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sn = y + (z * y) * (-1.66666666666666324348e-01 + z * rs);
  const double rc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 +
                                   z * (-2.75573143513906633035e-07 +
                                        z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + z * rc);
  // (cos, sin)(y + k pi/2)
  const double c = (k & 1) ? sn : cs, s_ = (k & 1) ? cs : sn;
  *cs_out = ((k + 1) & 2) ? -c : c;
  const double sa = (k & 2) ? -s_ : s_;
  *sn_out = x < 0.0 ? -sa : sa;
}

// (c, s) <- the same angle advanced by the angle whose cosine / sine are (cd, sd)
__device__ __forceinline__ void kwy_rotate(double &c, double &s, double cd, double sd) {
  const double c2 = c * cd - s * sd;
  s = s * cd + c * sd;
  c = c2;
}

// ------------------------------------------------------------ block reductions
// Cross-lane moves through the DPP path of the vector ALU (a few cycles) rather than
// ds_bpermute (__shfl_*: an LDS round trip per step, which is what the reductions and scans
// between two barriers used to spend most of their time on).  CTRL: 0x100+n row_shl:n (lane i
// reads lane i+n of its 16-lane row), 0x110+n row_shr:n, 0x142 row_bcast:15, 0x143 row_bcast:31.
// Lanes without a source (row boundary, masked row) read 0.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t kwy_dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double kwy_dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kwy_readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// sum over the wavefront; every lane gets the result
__device__ __forceinline__ double kwy_wave_sum(double v) {
  v += kwy_dpp_f64<0x101>(v);
  v += kwy_dpp_f64<0x102>(v);
  v += kwy_dpp_f64<0x104>(v);
  v += kwy_dpp_f64<0x108>(v);  // lane 0 of every row now holds its row's sum
  return (kwy_readlane_f64(v, 0) + kwy_readlane_f64(v, 16)) + (kwy_readlane_f64(v, 32) + kwy_readlane_f64(v, 48));
}

// minimum / maximum over the wavefront (every lane gets the result); lanes without a DPP source keep their own value
template <int CTRL>
__device__ __forceinline__ double kwy_dpp_keep_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kwy_wave_min_f64(double v) {
  v = fmin(v, kwy_dpp_keep_f64<0x101>(v));
  v = fmin(v, kwy_dpp_keep_f64<0x102>(v));
  v = fmin(v, kwy_dpp_keep_f64<0x104>(v));
  v = fmin(v, kwy_dpp_keep_f64<0x108>(v));  // lane 0 of every row now holds its row's minimum
  return fmin(fmin(kwy_readlane_f64(v, 0), kwy_readlane_f64(v, 16)), fmin(kwy_readlane_f64(v, 32), kwy_readlane_f64(v, 48)));
}
__device__ __forceinline__ double kwy_wave_max_f64(double v) {
  v = fmax(v, kwy_dpp_keep_f64<0x101>(v));
  v = fmax(v, kwy_dpp_keep_f64<0x102>(v));
  v = fmax(v, kwy_dpp_keep_f64<0x104>(v));
  v = fmax(v, kwy_dpp_keep_f64<0x108>(v));
  return fmax(fmax(kwy_readlane_f64(v, 0), kwy_readlane_f64(v, 16)), fmax(kwy_readlane_f64(v, 32), kwy_readlane_f64(v, 48)));
}

// inclusive prefix sums over the wavefront
__device__ __forceinline__ uint32_t kwy_wave_scan_u32(uint32_t v) {
  v += kwy_dpp_u32<0x111>(v);
  v += kwy_dpp_u32<0x112>(v);
  v += kwy_dpp_u32<0x114>(v);
  v += kwy_dpp_u32<0x118>(v);
  v += kwy_dpp_u32<0x142, 0xa>(v);  // rows 1 and 3 += last lane of the row before
  v += kwy_dpp_u32<0x143, 0xc>(v);  // rows 2 and 3 += lane 31
  return v;
}
__device__ __forceinline__ double kwy_wave_scan_f64(double v) {
  v += kwy_dpp_f64<0x111>(v);
  v += kwy_dpp_f64<0x112>(v);
  v += kwy_dpp_f64<0x114>(v);
  v += kwy_dpp_f64<0x118>(v);
  v += kwy_dpp_f64<0x142, 0xa>(v);
  v += kwy_dpp_f64<0x143, 0xc>(v);
  return v;
}

// sum over the whole block of NT threads; every thread gets the result. red: >= NT/64 doubles.
template <int NT = KWY_THREADS>
__device__ __forceinline__ double kwy_block_sum(double v, double *red) {
  v = kwy_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) s += red[i];
  return s;
}

// in-place inclusive prefix sum of buf[0..L) (LDS).  Thread t owns the
// contiguous chunk [t*chunk, (t+1)*chunk).  tot: NT doubles of LDS.
template <int NT = KWY_THREADS>
__device__ inline void kwy_block_cumsum(double *buf, int L, double *tot) {
  constexpr int PER = NT / 64;  // chunk totals scanned per lane of wave 0
  const int t = threadIdx.x;
  const int chunk = (L + NT - 1) / NT;
  const int b0 = t * chunk;
  const int b1 = min(L, b0 + chunk);
  double run = 0.0;
  for (int i = b0; i < b1; ++i) { run += buf[i]; buf[i] = run; }
  tot[t] = run;
  __syncthreads();
  if (t < 64) {  // wave 0 scans the NT chunk totals, PER per lane
    double a[PER];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { acc += tot[PER * t + q]; a[q] = acc; }
    const double inc = kwy_wave_scan_f64(acc);
    const double excl = inc - acc;  // sum of the lanes before this one
    tot[PER * t] = excl;
#pragma unroll
    for (int q = 1; q < PER; ++q) tot[PER * t + q] = excl + a[q - 1];
  }
  __syncthreads();
  const double off = tot[t];
  if (t > 0)
    for (int i = b0; i < b1; ++i) buf[i] += off;
  __syncthreads();
}

// The same prefix sum of values that are COMPUTED (value(i), i < L), left in buf: the thread's chunk stays in registers
// between its running sum and the addition of the offset, so the values are never stored, read back, summed in place
// and read again -- 2 LDS accesses per element instead of 6 and one barrier fewer than fill + kwy_block_cumsum.
// Same additions in the same order (bit-identical).  CH: compile-time bound of the chunk ceil(L / NT).
template <int NT, int CH, class F>
__device__ inline void kwy_block_cumsum_of(F value, double *buf, int L, double *tot) {
  constexpr int PER = NT / 64;
  const int t = threadIdx.x;
  const int chunk = (L + NT - 1) / NT;
  const int b0 = t * chunk;
  const int b1 = min(L, b0 + chunk);
  double r[CH];
  double run = 0.0;
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    r[q] = 0.0;
    if (q < chunk && b0 + q < b1) { run += value(b0 + q); r[q] = run; }
  }
  tot[t] = run;
  __syncthreads();
  if (t < 64) {  // wave 0 scans the NT chunk totals, PER per lane
    double a[PER];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { acc += tot[PER * t + q]; a[q] = acc; }
    const double inc = kwy_wave_scan_f64(acc);
    const double excl = inc - acc;  // sum of the lanes before this one
    tot[PER * t] = excl;
#pragma unroll
    for (int q = 1; q < PER; ++q) tot[PER * t + q] = excl + a[q - 1];
  }
  __syncthreads();
  const double off = tot[t];
#pragma unroll
  for (int q = 0; q < CH; ++q)
    if (q < chunk && b0 + q < b1) buf[b0 + q] = t > 0 ? r[q] + off : r[q];
  __syncthreads();
}

// offsets[i] = sum_{j<i} count(j) for i <= n, the counts given by a function of the index (evaluated twice: once for
// the chunk totals, once for the offsets) -- counting and scanning in ONE single-workgroup launch.  tot: NT uint64 of LDS.
template <int NT, class F>
__device__ inline void kwy_block_count_scan(F count, int64_t n, uint64_t *__restrict__ offsets, uint64_t *tot) {
  const int t = threadIdx.x;
  const int64_t chunk = (n + NT - 1) / NT;
  const int64_t b0 = t * chunk, b1 = min(n, b0 + chunk);
  uint64_t run = 0;
  for (int64_t i = b0; i < b1; ++i) run += count(i);
  tot[t] = run;
  __syncthreads();
  if (t < 64) {                      // wave 0: exclusive scan of the NT chunk totals, NT/64 per lane
    constexpr int PER = NT / 64;
    uint64_t a[PER], acc = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { a[q] = acc; acc += tot[PER * t + q]; }
    uint64_t inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t up = __shfl_up(inc, o);
      if (t >= o) inc += up;
    }
    const uint64_t excl = inc - acc;
#pragma unroll
    for (int q = 0; q < PER; ++q) tot[PER * t + q] = excl + a[q];
  }
  __syncthreads();
  run = tot[t];
  for (int64_t i = b0; i < b1; ++i) { offsets[i] = run; run += count(i); }
  if (b0 < n && b1 == n) offsets[n] = run;
  if (n == 0 && t == 0) offsets[0] = 0;
}

// sums of two values over the block in one exchange; red: >= 2*NT/64 doubles
template <int NT = KWY_THREADS>
__device__ __forceinline__ void kwy_block_sum2(double a, double b, double *red, double *ta, double *tb) {
  a = kwy_wave_sum(a);
  b = kwy_wave_sum(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[NT / 64 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  double sa = red[0], sb = red[NT / 64];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) { sa += red[i]; sb += red[NT / 64 + i]; }
  *ta = sa; *tb = sb;
}

// Sum of the m smallest of n non-negative doubles, and the sum of all of them, without sorting.  Thread t holds the
// IEEE bit patterns of elements t, t+NT, ... in key[] (~0 for slots beyond n); the patterns order like the values
// for x >= 0.  The m-th smallest value v* is found, then sum_small = sum(v < v*) + (m - #{v < v*}) * v*.
//   hist: KWY_SELECT_WORDS uint32 of LDS (16-byte aligned) that no thread touches any more when the call starts;
//   red: >= 2*NT/64 doubles.
// Two ways to v*:
//  (1) FROM THE TOP, one histogram: bin = distance of the key's high word from the block's largest high word in
//      quarter binades (256 bins = 64 binades below the maximum, everything lower in the last bin).  The bin that
//      holds the wanted rank -- counted from the largest value -- is left with a few dozen keys of a 2 049-bin
//      spectrum; they are listed in LDS and ranked against each other with full keys.  Only the maximum is reduced,
//      the keys stay as they are, three barriers + the ranking.  D4C asks for the 66th largest of 2 049: always here.
//  (2) GENERAL, MSB-first radix select (KWY_SELECT_BITS bits a round) on key - min(key), starting at the highest bit
//      in which the block's keys differ; two barriers per round, histogram and control words double-buffered; a
//      round that leaves <= 124 keys hands over to the same ranking.  Taken when (1) ends in its last bin or with a
//      crowded one (flat data, ties by the hundred, a rank far from the top).
// Digit width 8 (256 bins) is what the kernels use.  11 bits -- the whole exponent of a non-negative IEEE double in
// the first round -- was measured and is SLOWER (what the rounds save in barriers is lost clearing and scanning
// 2048 bins per round).  Wave-aggregated atomics (the lanes that share the first live lane's digit counted with one
// atomic) were measured too: on real group-delay spectra the ballots cost more issue slots than the same-address
// atomics they save (3 280 against 1 936 clocks for the nine slots of a round): KWY_SELECT_AGG=1 keeps that form.
#ifndef KWY_SELECT_BITS
#define KWY_SELECT_BITS 8
#endif
#define KWY_SELECT_BINS (1 << KWY_SELECT_BITS)
#ifndef KWY_SELECT_AGG
#define KWY_SELECT_AGG 0
#endif
#define KWY_SELECT_WORDS(NT) (2 * KWY_SELECT_BINS + 32)
#define KWY_SELECT_LIST (KWY_SELECT_BINS / 2 - 4)         // keys the ranking takes (the other histogram's words)

__device__ __forceinline__ uint32_t kwy_wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x101, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x102, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xf, 0xf, false));
  return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// wavefront 0: the bin of h[0 .. BINS) that holds rank kk (1-based, bins taken in index order) -> ctl[0] = bin,
// ctl[1] = rank inside it, ctl[2] = its population, ctl[3] = keys in the bins before it
__device__ __forceinline__ void kwy_select_scan(const uint32_t *h, int kk, uint32_t *ctl, int lane) {
  constexpr int PERLANE = KWY_SELECT_BINS / 64;           // bins scanned per lane
  static_assert(PERLANE % 4 == 0 && PERLANE >= 4, "bins per lane must be a multiple of 4");
  // lane l owns bins [PERLANE l, PERLANE (l + 1)): their total first, then only the lane whose range holds the
  // wanted rank walks its bins again (the counts are not kept in registers)
  const uint4 *hp = (const uint4 *)h + lane * (PERLANE / 4);
  uint32_t tot = 0;
#pragma unroll
  for (int q = 0; q < PERLANE / 4; ++q) {
    const uint4 c4 = hp[q];
    tot += (c4.x + c4.y) + (c4.z + c4.w);
  }
  uint32_t before = kwy_wave_scan_u32(tot) - tot;
  if ((uint32_t)kk > before && (uint32_t)kk <= before + tot) {
    const uint32_t *hb = h + PERLANE * lane;
    for (int q = 0; q < PERLANE; ++q) {
      const uint32_t c = hb[q];
      if ((uint32_t)kk <= before + c) {
        ctl[0] = PERLANE * lane + q;
        ctl[1] = (uint32_t)kk - before;
        ctl[2] = c;
        ctl[3] = before;
        break;
      }
      before += c;
    }
  }
}

// the listed keys list[0 .. pop) ranked against each other (pop <= KWY_SELECT_LIST; four entries behind the list
// must be readable).  DESCENDING: the key with `rank - 1` listed keys in front of it when sorted from the largest.
// Thread tid < pop looks at entry tid; the one that holds the wanted key publishes it with out_count[0] = the number
// of listed keys in front of it (strictly) and out_count[1] = the listed keys equal to it.  Ascending order: the same
// with the comparison turned round.
template <bool DESCENDING>
__device__ __forceinline__ void kwy_select_rank(const unsigned long long *list, int pop, int rank, int tid,
                                                unsigned long long *out_key, uint32_t *out_count) {
  if (tid < pop) {
    const unsigned long long mine = list[tid];
    int front = 0, tie_front = 0, ties = 0;
    for (int j = 0; j < pop; j += 4) {                       // four keys a trip: two 16-byte reads in flight
      const ulonglong2 o01 = *(const ulonglong2 *)(list + j), o23 = *(const ulonglong2 *)(list + j + 2);
      const unsigned long long o[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool in = j + q < pop;
        front += in & (DESCENDING ? o[q] > mine : o[q] < mine);
        const bool eq = in & (o[q] == mine);
        ties += eq;
        tie_front += eq & (j + q < tid);
      }
    }
    if (front + tie_front == rank - 1) { *out_key = mine; out_count[0] = (uint32_t)front; out_count[1] = (uint32_t)ties; }
  }
}

template <int RMAX, int NT = KWY_THREADS>
__device__ inline void kwy_block_smallest_sum(unsigned long long (&key)[RMAX], int n, int m,
                                              uint32_t *hist, double *red, double *sum_small,
                                              double *sum_all) {
  constexpr int BITS = KWY_SELECT_BITS, BINS = 1 << BITS;
  constexpr int TOP_SHIFT = 18;                           // high word: 20 mantissa bits -> quarter binades
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t *ctlb = hist + 2 * BINS;  // 2 x {digit, rank, population, before}; [8..9] one 64-bit key; [10] list
                                     // length; [11..12] two counts; [16 ..] one word per wavefront
  unsigned long long *k64 = (unsigned long long *)(ctlb + 8);
  unsigned long long *list = (unsigned long long *)(hist + BINS);
  unsigned long long vkey;           // bit pattern of v*
  int copies;                        // how many keys equal to v* belong to the m smallest

  // ---- (1) from the top
  uint32_t hmax = 0;
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (tid + NT * r < n) hmax = max(hmax, (uint32_t)(key[r] >> 32));
  hmax = kwy_wave_max_u32(hmax);
  if (lane == 0) ctlb[16 + wv] = hmax;
  for (int i = tid; i < BINS; i += NT) hist[i] = 0;
  if (tid == 0) ctlb[10] = 0;
  __syncthreads();
  hmax = ctlb[16];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) hmax = max(hmax, ctlb[16 + i]);
  uint32_t dg[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    dg[r] = BINS;                                         // (no bin: slot beyond n)
    if (NT * r + (tid & ~63) >= n) continue;              // nothing in this slot for the whole wavefront
    if (tid + NT * r < n) {
      dg[r] = min((hmax - (uint32_t)(key[r] >> 32)) >> TOP_SHIFT, (uint32_t)(BINS - 1));
      atomicAdd(&hist[dg[r]], 1u);
    }
  }
  __syncthreads();
  if (wv == 0) kwy_select_scan(hist, n - m + 1, ctlb, lane);
  __syncthreads();
  const uint32_t chosen = ctlb[0], pop = ctlb[2];
  if (chosen < (uint32_t)(BINS - 1) && pop <= (uint32_t)KWY_SELECT_LIST) {
    const int rank = (int)ctlb[1], above = (int)ctlb[3];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (dg[r] == chosen) list[atomicAdd(&ctlb[10], 1u)] = key[r];
    __syncthreads();
    kwy_select_rank<true>(list, (int)pop, rank, tid, k64, &ctlb[11]);
    __syncthreads();
    vkey = *k64;
    copies = m - (n - above - (int)ctlb[11] - (int)ctlb[12]);   // keys >= v*: above + listed in front + ties
  } else {
    // ---- (2) general: the block's smallest and largest key, relative keys, radix rounds
    double lo = INFINITY, hi = 0.0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      if (tid + NT * r < n) {
        const double x = __longlong_as_double((long long)key[r]);
        lo = fmin(lo, x); hi = fmax(hi, x);
      }
    }
    lo = kwy_wave_min_f64(lo);
    hi = kwy_wave_max_f64(hi);
    if (lane == 0) { red[wv] = lo; red[NT / 64 + wv] = hi; }
    for (int i = tid; i < BINS; i += NT) hist[i] = 0;
    if (tid == 0) ctlb[10] = 0;
    __syncthreads();
    lo = red[0]; hi = red[NT / 64];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) { lo = fmin(lo, red[i]); hi = fmax(hi, red[NT / 64 + i]); }
    const unsigned long long kmin = (unsigned long long)__double_as_longlong(lo);
    const unsigned long long range = (unsigned long long)__double_as_longlong(hi) - kmin;
    const int top0 = range ? 64 - __clzll((long long)range) : 0;     // one past the highest differing bit
    const int rounds = (top0 + BITS - 1) / BITS;                    // 0: all keys equal
#pragma unroll
    for (int r = 0; r < RMAX; ++r) key[r] -= kmin;                  // (slots beyond n are never looked at)
    unsigned long long prefix = 0ull;
    int kk = m;  // 1-based rank of the wanted element among the still-matching keys
    bool alive[RMAX];  // key still carries the prefix found so far
#pragma unroll
    for (int r = 0; r < RMAX; ++r) alive[r] = tid + NT * r < n;
    for (int round = 0; round < rounds; ++round) {
      // digits are taken from bit top0 - 1 downwards; the last one may be narrower
      const int top = top0 - BITS * round;                  // one past the digit's highest bit
      const int shift = top - BITS > 0 ? top - BITS : 0;
      const unsigned long long mask = (1ull << (top - shift)) - 1ull;
      uint32_t *h = hist + (round & 1) * BINS, *hn = hist + ((round + 1) & 1) * BINS;
      uint32_t *ctl = ctlb + (round & 1) * 4;
      for (int i = tid; i < BINS; i += NT) hn[i] = 0;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        if (NT * r + (tid & ~63) >= n) continue;  // nothing in this slot for the whole wavefront
        const int d = (int)((key[r] >> shift) & mask);
#if KWY_SELECT_AGG
        const unsigned long long mm = __ballot(alive[r]);
        if (mm != 0ull) {
          const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
          const int d0 = __builtin_amdgcn_readlane(d, leader);
          const bool same = alive[r] && d == d0;
          const unsigned long long ms = __ballot(same);
          if (lane == leader) atomicAdd(&h[d0], (uint32_t)__popcll(ms));
          if (alive[r] && !same) atomicAdd(&h[d], 1u);
        }
#else
        if (alive[r]) atomicAdd(&h[d], 1u);
#endif
      }
      __syncthreads();
      if (wv == 0) kwy_select_scan(h, kk, ctl, lane);
      __syncthreads();
      const int chosen_d = (int)ctl[0];
      prefix |= (unsigned long long)chosen_d << shift;
      kk = (int)ctl[1];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) alive[r] = alive[r] && (int)((key[r] >> shift) & mask) == chosen_d;
      const uint32_t left = ctl[2];
      if (left == 1u && round < rounds - 1) {
        // exactly one key carries this prefix: it IS the wanted element, skip the remaining rounds
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
          if (alive[r]) *k64 = key[r];
        __syncthreads();
        prefix = *k64;
        kk = 1;
        break;
      }
      if (left <= (uint32_t)KWY_SELECT_LIST && round < rounds - 1) {
        // a handful of keys carry this prefix: listed in the other histogram's words and ranked against each other
        unsigned long long *lst = (unsigned long long *)hn;
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
          if (alive[r]) lst[atomicAdd(&ctlb[10], 1u)] = key[r];
        __syncthreads();
        kwy_select_rank<false>(lst, (int)left, kk, tid, k64, &ctlb[11]);
        __syncthreads();
        prefix = *k64;
        kk -= (int)ctlb[11];                                // copies of v* among the m smallest
        break;
      }
    }
    // (all keys equal: no round ran, prefix = 0 = every relative key, kk = m of them)
    vkey = prefix + kmin;
    copies = kk;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) key[r] += kmin;
  }
  const double vstar = __longlong_as_double((long long)vkey);
  double s_less = 0.0, s_all = 0.0;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (tid + NT * r < n) {
      const double x = __longlong_as_double((long long)key[r]);
      s_all += x;
      if (key[r] < vkey) s_less += x;
    }
  }
  double t_less, t_all;
  kwy_block_sum2<NT>(s_less, s_all, red, &t_less, &t_all);
  *sum_small = t_less + (double)copies * vstar;
  *sum_all = t_all;
}

// ------------------------------------------------------------------- LDS FFTs
struct __attribute__((aligned(16))) kwy_c { double x, y; };  // complex double (16 B, LDS b128 accesses)

__device__ __forceinline__ kwy_c cadd(kwy_c a, kwy_c b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ kwy_c csub(kwy_c a, kwy_c b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ kwy_c cmul(kwy_c a, kwy_c b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// The same product with explicit fused multiply-adds: 4 instructions instead of 6.  The library is compiled with
// -ffp-contract=off because several kernels reproduce serial CPU roundings exactly (phase accumulation, DTW costs,
// numpy's accept / reject); the FFT passes are not among them -- their results are compared at 1e-8 ... 1e-12 -- and
// the twiddle products are 40 % of a radix-8 pass.
__device__ __forceinline__ kwy_c cmulf(kwy_c a, kwy_c b) {
  return {__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x)};
}

// exp(-2 pi i (t + r*NT) / N) from base = exp(-2 pi i t / N) when NT = N/8: base times an 8th root of unity
__device__ __forceinline__ kwy_c kwy_tw_octant(kwy_c b, int r) {
  const double h = 0.70710678118654752440;
  switch (r & 7) {
    case 0: return b;
    case 1: return {h * (b.x + b.y), h * (b.y - b.x)};
    case 2: return {b.y, -b.x};
    case 3: return {h * (b.y - b.x), h * (-b.x - b.y)};
    case 4: return {-b.x, -b.y};
    case 5: return {h * (-b.x - b.y), h * (b.x - b.y)};
    case 6: return {-b.y, b.x};
    default: return {h * (b.x - b.y), h * (b.x + b.y)};
  }
}

// ------------------------------------------------- in-place radix-8 LDS FFT
// One buffer instead of the ping-pong pair above: every pass loads its operands
// into registers, all threads meet at a barrier, and the results go back to the
// same array (Stockham ordering, so the output is in natural order).  8 points
// per thread and pass: an H = 2048 point transform is 8*8*8*4 = 4 LDS round trips.
template <bool INV>
__device__ __forceinline__ kwy_c kwy_rot90(kwy_c a) {  // a * (-i) forward, a * (+i) inverse
  return INV ? kwy_c{-a.y, a.x} : kwy_c{a.y, -a.x};
}

template <bool INV>
__device__ __forceinline__ void kwy_dft8(kwy_c (&a)[8]) {
  const double h = 0.70710678118654752440;
  kwy_c t0 = cadd(a[0], a[4]), t1 = csub(a[0], a[4]);
  kwy_c t2 = cadd(a[2], a[6]), t3 = kwy_rot90<INV>(csub(a[2], a[6]));
  kwy_c t4 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]);
  kwy_c t6 = cadd(a[3], a[7]), t7 = csub(a[3], a[7]);
  kwy_c u0 = cadd(t0, t2), u1 = csub(t0, t2), u2 = cadd(t4, t6), u3 = kwy_rot90<INV>(csub(t4, t6));
  // c1 = t5 * w8, c3 = t7 * w8^3 with w8 = exp(-+ i pi/4)
  kwy_c c1 = INV ? kwy_c{h * (t5.x - t5.y), h * (t5.x + t5.y)} : kwy_c{h * (t5.x + t5.y), h * (t5.y - t5.x)};
  kwy_c c3 = INV ? kwy_c{h * (-t7.x - t7.y), h * (t7.x - t7.y)} : kwy_c{h * (t7.y - t7.x), h * (-t7.x - t7.y)};
  kwy_c v0 = cadd(t1, t3), v1 = csub(t1, t3), v2 = cadd(c1, c3), v3 = kwy_rot90<INV>(csub(c1, c3));
  a[0] = cadd(u0, u2); a[4] = csub(u0, u2); a[2] = cadd(u1, u3); a[6] = csub(u1, u3);
  a[1] = cadd(v0, v2); a[5] = csub(v0, v2); a[3] = cadd(v1, v3); a[7] = csub(v1, v3);
}

// Radix-8 pass with sub-transform stride S = 2^LOG2S over H points, in place.
// The first pass (S = 1) would store each thread's 8 results 128 B apart -- an
// 8-way conflict on ds_write_b128 -- so it stores to i ^ ((i >> 3) & 7) instead,
// and the second pass (S = 8) reads through the same permutation.
// tw: exp(-2 pi i k / H) for k < H/8 at least (the factors of one radix-8 pass are tw[ps], ps < H/8,
// and its 2nd .. 7th powers, which are formed by multiplication); it may live in LDS.
struct kwy_tw_table {   // pass factor from a table exp(-2 pi i k / H) (global or LDS)
  const kwy_c *tw;
  __device__ __forceinline__ kwy_c operator()(int ps) const { return tw[ps]; }
};
struct kwy_tw_reg {     // pass factor held by the thread itself (one butterfly per thread and pass)
  kwy_c w;
  // behind an optimisation barrier: its 2nd..7th powers are formed again in every pass instead of
  // being computed once per kernel and carried (spilled) across all transforms
  __device__ __forceinline__ kwy_c operator()(int) const {
    kwy_c v = w;
    asm volatile("" : "+v"(v.x), "+v"(v.y));
    return v;
  }
};

template <int LOG2H, int LOG2S, int NT, bool INV, class TW>
__device__ __forceinline__ void kwy_fft_pass8_core(kwy_c *z, TW tw) {
  constexpr int H = 1 << LOG2H, Q = H / 8, S = 1 << LOG2S;
  constexpr int IT = (Q + NT - 1) / NT;
  constexpr bool LAST = (LOG2S + 3 == LOG2H);
  static_assert(Q >= 8, "transform too short for the radix-8 kernel");
  kwy_c a[IT][8];
  const int tid = kwy_tid_opaque();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * NT;
    if (j < Q) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int idx = j + m * Q;
        a[it][m] = z[(LOG2S == 3) ? (idx ^ ((idx >> 3) & 7)) : idx];
      }
      kwy_dft8<INV>(a[it]);
      if (!LAST) {
        const int ps = (j >> LOG2S) << LOG2S;
        kwy_c w1 = tw(ps);
        if (INV) w1.y = -w1.y;
        const kwy_c w2 = cmulf(w1, w1), w4 = cmulf(w2, w2);
        const kwy_c w3 = cmulf(w1, w2), w5 = cmulf(w4, w1), w6 = cmulf(w4, w2);
        const kwy_c w7 = cmulf(w4, w3);
        a[it][1] = cmulf(w1, a[it][1]); a[it][2] = cmulf(w2, a[it][2]); a[it][3] = cmulf(w3, a[it][3]);
        a[it][4] = cmulf(w4, a[it][4]); a[it][5] = cmulf(w5, a[it][5]); a[it][6] = cmulf(w6, a[it][6]);
        a[it][7] = cmulf(w7, a[it][7]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * NT;
    if (j < Q) {
      const int q = j & (S - 1), p = j >> LOG2S;
      const int o = q + ((8 * p) << LOG2S);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (LOG2S == 0) z[8 * j + (m ^ (j & 7))] = a[it][m];
        else z[o + (m << LOG2S)] = a[it][m];
      }
    }
  }
  __syncthreads();
}

template <int LOG2H, int LOG2S, int NT, bool INV>
__device__ __forceinline__ void kwy_fft_pass8(kwy_c *z, const kwy_c *__restrict__ tw) {
  kwy_fft_pass8_core<LOG2H, LOG2S, NT, INV>(z, kwy_tw_table{tw});
}

// First radix-8 pass (S = 1) of a transform whose input is zero except for x[0 .. H/8]: thread j
// supplies x[j] in a0, thread 0 also x[H/8] in a1.  The butterfly degenerates to a copy (every
// output of thread j is x[j] times its twiddle), so nothing is read from LDS and no barrier is
// needed before the stores -- as long as the buffer itself is free.  Ends with a barrier.
template <int LOG2H, int NT, bool INV, class TW>
__device__ __forceinline__ void kwy_fft_pass8_first_sparse_core(kwy_c *z, TW tw, kwy_c a0, kwy_c a1) {
  constexpr int H = 1 << LOG2H, Q = H / 8;
  static_assert(Q <= NT, "one butterfly per thread");
  const int j = kwy_tid_opaque();
  if (j < Q) {
    kwy_c a[8];
    if (j == 0) {
      a[0] = a0; a[1] = a1;
#pragma unroll
      for (int m = 2; m < 8; ++m) a[m] = {0.0, 0.0};
      kwy_dft8<INV>(a);
    } else {
      kwy_c w1 = tw(j);
      if (INV) w1.y = -w1.y;
      const kwy_c w2 = cmulf(w1, w1), w4 = cmulf(w2, w2);
      const kwy_c w3 = cmulf(w1, w2), w5 = cmulf(w4, w1), w6 = cmulf(w4, w2);
      const kwy_c w7 = cmulf(w4, w3);
      a[0] = a0; a[1] = cmulf(w1, a0); a[2] = cmulf(w2, a0); a[3] = cmulf(w3, a0);
      a[4] = cmulf(w4, a0); a[5] = cmulf(w5, a0); a[6] = cmulf(w6, a0); a[7] = cmulf(w7, a0);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) z[8 * j + (m ^ (j & 7))] = a[m];
  }
  __syncthreads();
}

template <int LOG2H, int NT, bool INV>
__device__ __forceinline__ void kwy_fft_pass8_first_sparse(kwy_c *z, const kwy_c *__restrict__ tw, kwy_c a0, kwy_c a1) {
  kwy_fft_pass8_first_sparse_core<LOG2H, NT, INV>(z, kwy_tw_table{tw}, a0, a1);
}

template <int LOG2H, int TAIL, int NT, bool INV>
__device__ __forceinline__ void kwy_fft_tail(kwy_c *z);

// the remaining passes after kwy_fft_pass8_first_sparse
template <int LOG2H, int NT, bool INV>
__device__ inline void kwy_fft_inplace_rest(kwy_c *z, const kwy_c *__restrict__ tw) {
  kwy_fft_pass8<LOG2H, 3, NT, INV>(z, tw);
  kwy_fft_pass8<LOG2H, 6, NT, INV>(z, tw);
  if constexpr (LOG2H == 12) kwy_fft_pass8<LOG2H, 9, NT, INV>(z, tw);
  if constexpr (LOG2H == 11) kwy_fft_tail<LOG2H, 2, NT, INV>(z);
  if constexpr (LOG2H == 10) kwy_fft_tail<LOG2H, 1, NT, INV>(z);
}

// closing radix-4 (TAIL = 2) or radix-2 (TAIL = 1) pass: sub-transform stride H/4 resp. H/2, no twiddles
template <int LOG2H, int TAIL, int NT, bool INV>
__device__ __forceinline__ void kwy_fft_tail(kwy_c *z) {
  constexpr int H = 1 << LOG2H, R = 1 << TAIL, Q = H / R;
  constexpr int IT = (Q + NT - 1) / NT;
  kwy_c a[IT][R];
  const int tid = kwy_tid_opaque();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * NT;
    if (j < Q) {
#pragma unroll
      for (int m = 0; m < R; ++m) a[it][m] = z[j + m * Q];
      if constexpr (TAIL == 2) {
        kwy_c apc = cadd(a[it][0], a[it][2]), amc = csub(a[it][0], a[it][2]);
        kwy_c bpd = cadd(a[it][1], a[it][3]), jb = kwy_rot90<INV>(csub(a[it][1], a[it][3]));
        a[it][0] = cadd(apc, bpd); a[it][1] = cadd(amc, jb);
        a[it][2] = csub(apc, bpd); a[it][3] = csub(amc, jb);
      } else {
        kwy_c s0 = cadd(a[it][0], a[it][1]), s1 = csub(a[it][0], a[it][1]);
        a[it][0] = s0; a[it][1] = s1;
      }
      // the closing pass writes exactly the locations it read: no barrier in between
#pragma unroll
      for (int m = 0; m < R; ++m) z[j + m * Q] = a[it][m];
    }
  }
  __syncthreads();
}

// In-place complex FFT of H = 2^LOG2H (512 .. 4096) points in LDS.  The caller
// has a barrier between filling z and this call; ends with a barrier.
// tw: exp(-2 pi i k / H), k < H/8 (global or LDS).  Unnormalised in both directions.
template <int LOG2H, int NT, bool INV>
__device__ inline void kwy_fft_inplace(kwy_c *z, const kwy_c *__restrict__ tw) {
  static_assert(LOG2H >= 8 && LOG2H <= 12, "unsupported in-place FFT length");
  kwy_fft_pass8<LOG2H, 0, NT, INV>(z, tw);
  kwy_fft_pass8<LOG2H, 3, NT, INV>(z, tw);
  if constexpr (LOG2H >= 9) kwy_fft_pass8<LOG2H, 6, NT, INV>(z, tw);
  if constexpr (LOG2H == 12) kwy_fft_pass8<LOG2H, 9, NT, INV>(z, tw);
  if constexpr (LOG2H % 3 != 0) kwy_fft_tail<LOG2H, LOG2H % 3, NT, INV>(z);
}

// The same transform with the pass factors held by the threads: when a pass has one butterfly
// per thread (H/8 <= NT), the factor of pass p is the thread constant
// w[p] = exp(-2 pi i ((tid >> 3p) << 3p) / H)  -- no table at all (kwy_fft_thread_twiddles).
template <int LOG2H, int NT>
__device__ __forceinline__ void kwy_fft_thread_twiddles(const kwy_c *__restrict__ twH, kwy_c (&w)[4]) {
  static_assert((1 << LOG2H) / 8 <= NT, "one butterfly per thread and pass");
  const int j = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ps = (j >> (3 * p)) << (3 * p);
    w[p] = (3 * p + 3 < LOG2H + 0 && ps < (1 << LOG2H) / 8) ? twH[ps] : kwy_c{1.0, 0.0};
  }
}
template <int LOG2H, int NT, bool INV>
__device__ inline void kwy_fft_inplace_rest_w(kwy_c *z, const kwy_c (&w)[4]) {
  kwy_fft_pass8_core<LOG2H, 3, NT, INV>(z, kwy_tw_reg{w[1]});
  if constexpr (LOG2H >= 9) kwy_fft_pass8_core<LOG2H, 6, NT, INV>(z, kwy_tw_reg{w[2]});
  if constexpr (LOG2H == 12) kwy_fft_pass8_core<LOG2H, 9, NT, INV>(z, kwy_tw_reg{w[3]});
  if constexpr (LOG2H % 3 != 0) kwy_fft_tail<LOG2H, LOG2H % 3, NT, INV>(z);
}
template <int LOG2H, int NT, bool INV>
__device__ inline void kwy_fft_inplace_w(kwy_c *z, const kwy_c (&w)[4]) {
  static_assert(LOG2H >= 8 && LOG2H <= 12, "unsupported in-place FFT length");
  kwy_fft_pass8_core<LOG2H, 0, NT, INV>(z, kwy_tw_reg{w[0]});
  kwy_fft_inplace_rest_w<LOG2H, NT, INV>(z, w);
}

// exp(-2 pi i (t + r*NT) / N) from base = exp(-2 pi i t / N): base times a 16th root of unity,
// idx16 = r * 16 NT / N.  Even indices go through the exact 8th-root form.
__device__ __forceinline__ kwy_c kwy_tw_hex(kwy_c b, int idx16) {
  if (!(idx16 & 1)) return kwy_tw_octant(b, idx16 >> 1);
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;  // cos, sin of pi/8
  double wr, wi;  // exp(-i pi idx16 / 8)
  switch (idx16 & 15) {
    case 1: wr = c1; wi = -s1; break;
    case 3: wr = s1; wi = -c1; break;
    case 5: wr = -s1; wi = -c1; break;
    case 7: wr = -c1; wi = -s1; break;
    case 9: wr = -c1; wi = s1; break;
    case 11: wr = -s1; wi = c1; break;
    case 13: wr = s1; wi = c1; break;
    default: wr = c1; wi = s1; break;
  }
  return {__builtin_fma(b.x, wr, -(b.y * wi)), __builtin_fma(b.x, wi, b.y * wr)};
}

// Real transforms around it, in place in a buffer of H+1 complex.
// kwy_rfft_inplace: z holds N = 2H reals (viewed as H packed complex); on return z[0..H] are the
// bins X[0..H].  Each thread combines "its" pairs (k, H-k) -- both results come from the same two
// packed values -- so no barrier separates the loads from the stores.  twb = exp(-2 pi i tid / N):
// for NT >= N/8 the pair twiddles are twb times 8th roots of unity, else they are read from twN[].
// Ends with a barrier.
template <int LOG2H, int NT>
__device__ inline void kwy_rfft_inplace(kwy_c *z, const kwy_c *__restrict__ tw, kwy_c twb,
                                   const kwy_c *__restrict__ twN = nullptr) {
  constexpr int H = 1 << LOG2H, N = 2 * H;
  constexpr int OCT = 8 * NT / N;  // 0: the workgroup is narrower than N/8, pair twiddles come from twN[]
  kwy_fft_inplace<LOG2H, NT, false>(z, tw);
  const int tid = kwy_tid_opaque();
#pragma unroll
  for (int r = 0; r * NT <= H / 2; ++r) {
    const int k = tid + NT * r;
    if (k > H / 2) continue;
    if (k == 0) {
      const kwy_c z0 = z[0];
      z[0] = {z0.x + z0.y, 0.0};
      z[H] = {z0.x - z0.y, 0.0};
    } else {
      const kwy_c w = (OCT >= 1) ? kwy_tw_octant(twb, OCT * r) : twN[k];
      const kwy_c A = z[k], Bc = z[H - k];
      // X[k] = E + O w,  X[H-k] = conj(E - O w)  with E = (A + conj B)/2, O = (A - conj B)/(2i)
      const double er = 0.5 * (A.x + Bc.x), ei = 0.5 * (A.y - Bc.y);
      const double dr = 0.5 * (A.x - Bc.x), di = 0.5 * (A.y + Bc.y);
      const double orr = di, oi = -dr;
      const double pr = __builtin_fma(orr, w.x, -(oi * w.y)), pi = __builtin_fma(orr, w.y, oi * w.x);
      z[k] = {er + pr, ei + pi};
      if (k != H - k) z[H - k] = {er - pr, -(ei - pi)};
    }
  }
  __syncthreads();
}

// kwy_irfft_inplace: z[0..H] holds X; on return the first N doubles of z are the signal times N
// (unnormalised: N times the true inverse).  Ends with a barrier.
template <int LOG2H, int NT>
__device__ inline void kwy_irfft_inplace(kwy_c *z, const kwy_c *__restrict__ tw, kwy_c twb,
                                   const kwy_c *__restrict__ twN = nullptr) {
  constexpr int H = 1 << LOG2H, N = 2 * H;
  constexpr int OCT = 8 * NT / N;  // 0: the workgroup is narrower than N/8, pair twiddles come from twN[]
  const int tid = kwy_tid_opaque();
  __syncthreads();
#pragma unroll
  for (int r = 0; r * NT <= H / 2; ++r) {
    const int k = tid + NT * r;
    if (k > H / 2) continue;
    if (k == 0) {
      // b[0] from a[0], a[H] (imaginary parts of the DC and Nyquist bins ignored)
      const double ar = z[0].x, br = z[H].x;
      z[0] = {ar + br, ar - br};
    } else {
      const kwy_c w = (OCT >= 1) ? kwy_tw_octant(twb, OCT * r) : twN[k];  // exp(-2 pi i k / N); the inverse conjugates it
      const kwy_c A = z[k], Bc = z[H - k];
      // b[k]: (a, conj b) = (A, conj Bc);  b[H-k]: (Bc, conj A), twiddle -conj(w)
      const double er = A.x + Bc.x, ei = A.y - Bc.y;
      const double dr = A.x - Bc.x, di = A.y + Bc.y;
      const double wr = w.x, wi = -w.y;
      const double orr = __builtin_fma(dr, wr, -(di * wi)), oi = __builtin_fma(dr, wi, di * wr);
      z[k] = {er - oi, ei + orr};
      if (k != H - k) z[H - k] = {er + oi, -(ei - orr)};
    }
  }
  __syncthreads();
  kwy_fft_inplace<LOG2H, NT, true>(z, tw);
}

// Bin k (0 <= k <= H) of the real FFT of the N = 2H reals whose packed
// half-length transform sits in z; twN: exp(-2 pi i k / N).  Read-only on z.
template <int LOG2H>
__device__ __forceinline__ kwy_c kwy_rfft_bin_w(const kwy_c *z, int k, kwy_c w) {  // w = exp(-2 pi i k / N)
  constexpr int H = 1 << LOG2H;
  if (k == 0) return {z[0].x + z[0].y, 0.0};
  if (k == H) return {z[0].x - z[0].y, 0.0};
  const kwy_c A = z[k];
  const kwy_c B = {z[H - k].x, -z[H - k].y};
  const double er = 0.5 * (A.x + B.x), ei = 0.5 * (A.y + B.y);
  const double dr = 0.5 * (A.x - B.x), di = 0.5 * (A.y - B.y);
  const double orr = di, oi = -dr;
  return {er + (orr * w.x - oi * w.y), ei + (orr * w.y + oi * w.x)};
}
// The same bin times TWO (the four halvings of the even / odd split are left out): for consumers that only form
// ratios of quadratic expressions of the bins -- D4C's centroid over power spectrum, band energy ratios, LoveTrain's
// cumulative power ratio -- the factor cancels exactly (a power of two), and every bin saves four multiplications.
template <int LOG2H>
__device__ __forceinline__ kwy_c kwy_rfft_bin2_w(const kwy_c *z, int k, kwy_c w) {  // w = exp(-2 pi i k / N)
  constexpr int H = 1 << LOG2H;
  if (k == 0) return {2.0 * (z[0].x + z[0].y), 0.0};
  if (k == H) return {2.0 * (z[0].x - z[0].y), 0.0};
  const kwy_c A = z[k];
  const kwy_c B = {z[H - k].x, -z[H - k].y};
  const double er = A.x + B.x, ei = A.y + B.y;
  const double dr = A.x - B.x, di = A.y - B.y;
  // 2 X = (er, ei) + (di, -dr) w, as two fused multiply-adds per component (see cmulf)
  return {__builtin_fma(dr, w.y, __builtin_fma(di, w.x, er)), __builtin_fma(-dr, w.x, __builtin_fma(di, w.y, ei))};
}
// |2 X[k]|^2 and |2 X[H - k]|^2 together (0 <= k <= H/2): the two bins are E + O w and conj(E - O w) of the same two
// packed points -- one pair of LDS reads, one twiddle and one complex product for two powers
template <int LOG2H>
__device__ __forceinline__ void kwy_rfft_pair_power2_w(const kwy_c *z, int k, kwy_c w, double *pk, double *pm) {
  constexpr int H = 1 << LOG2H;
  if (k == 0) {
    const double a = 2.0 * (z[0].x + z[0].y), b = 2.0 * (z[0].x - z[0].y);
    *pk = a * a; *pm = b * b;
    return;
  }
  const kwy_c A = z[k], Bz = z[H - k];
  const double er = A.x + Bz.x, ei = A.y - Bz.y;
  const double dr = A.x - Bz.x, di = A.y + Bz.y;
  const double pr = __builtin_fma(dr, w.y, di * w.x), pi = __builtin_fma(-dr, w.x, di * w.y);
  const double xr = er + pr, xi = ei + pi, yr = er - pr, yi = ei - pi;
  *pk = __builtin_fma(xr, xr, xi * xi);
  *pm = __builtin_fma(yr, yr, yi * yi);
}
template <int LOG2H>
__device__ __forceinline__ kwy_c kwy_rfft_bin(const kwy_c *z, int k, const kwy_c *__restrict__ twN) {
  return kwy_rfft_bin_w<LOG2H>(z, k, twN[k & ((2 << LOG2H) - 1)]);
}

__device__ __forceinline__ int kwy_matlab_round(double x) {
  return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5);
}
