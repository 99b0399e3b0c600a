// kwy_mcep.hip -- spectral envelope <-> mel-cepstrum on gfx950.
//
// Replaces pysptk.sp2mc / pysptk.mc2sp (reference call sites
// kwiiyatta/vocoder/mcep.py:71 and :65; pysptk 0.1.16 / SPTK `freqt`):
//
//   sp2mc: c = irfft(log P); c[0] /= 2; mc = freqt(c, order, alpha)
//   mc2sp: c = freqt(mc, fftlen/2, -alpha); c[0] *= 2; mirror; exp(real(rfft(c)))
//
// SPTK's freqt is a linear recursion, so it is applied as a small dense matrix (built once per (alpha, sizes) on the
// host with the same recursion and cached in HBM).
//   sp2mc, power-of-two transforms: one workgroup per MC_FR frames does the inverse FFT of the log-spectrum in LDS
//          and the matrix product from L2-resident coefficients; other lengths: the whole map as one dense K x 25
//          matrix (k_sp2mc_dense).
//   mc2sp, every length: the whole map as a dense 25 x K matrix applied on the matrix cores (k_mc2sp_mfma,
//          v_mfma_f64_16x16x4_f64), exp on the accumulators.  It replaced the LDS-FFT form in round 2 (15 us against
//          35 us for 2201 frames at K = 1025); the same formulation of sp2mc was measured slower than its FFT form
//          twice (73 us with the log-spectra staged in LDS and a barrier pair per 256 bins, 46 us with eight
//          independent wavefronts per 16 frames and no staging, against 37 us).
// Both stream each frame once: sp2mc reads K*8 B and writes (order+1)*8 B per frame; mc2sp the reverse.
#include <math.h>

#include <map>
#include <mutex>
#include <string>

#include <vector>

#include "kwy_internal.hpp"

#define MC_MAX_ORDER 63  // order+1 <= 64 coefficients

// ---- host: freqt as a matrix ---------------------------------------------------
// One freqt step with zero input: g <- B g  (SPTK freqt inner loop with c1[-i] = 0)
static void freqt_step0(std::vector<double> &g, std::vector<double> &d, int m2, double a) {
  const double b = 1 - a * a;
  if (0 <= m2) { d[0] = g[0]; g[0] = a * d[0]; }
  if (1 <= m2) { d[1] = g[1]; g[1] = b * d[0] + a * d[1]; }
  for (int j = 2; j <= m2; ++j) { d[j] = g[j]; g[j] = d[j - 1] + a * (d[j] - g[j - 1]); }
}

// F[n][j] (n < ncols, j <= m2): response of output j to a unit input at index n.
// Input n is injected and then transformed n more times, so F[n] = B^n e0.
static void freqt_matrix(int ncols, int m2, double a, std::vector<double> &F) {
  F.assign((size_t)ncols * (m2 + 1), 0.0);
  std::vector<double> g(m2 + 1, 0.0), d(m2 + 1, 0.0);
  g[0] = 1.0;
  for (int n = 0; n < ncols; ++n) {
    for (int j = 0; j <= m2; ++j) F[(size_t)n * (m2 + 1) + j] = g[j];
    freqt_step0(g, d, m2, a);
  }
}

// ---- sp2mc -----------------------------------------------------------------------
// F: [ncut][MC_STRIDE] with MC_STRIDE = 64 doubles per cepstral index.
// A workgroup converts MC_FR frames: the cepstra are formed one after the other in the one FFT
// buffer (in-place inverse transform) and parked in LDS, then the freqt matrix -- 360 KB, read from
// L2 -- is walked ONCE for all of them.
#define MC_STRIDE 64
#define MC_FR 4
template <int LOG2N>
__global__ __launch_bounds__(KWY_THREADS) void k_sp2mc(const double *__restrict__ sp, int64_t T, int order,
                                                      const double *__restrict__ F, int ncut,
                                                      const kwy_c *__restrict__ twH,
                                                      const kwy_c *__restrict__ twN,
                                                      double *__restrict__ mc) {
  constexpr int N = 1 << LOG2N, H = N / 2, K = H + 1;
  constexpr int TWL = (H / 8 > 1) ? H / 8 : 1;
  extern __shared__ double smem[];
  kwy_c *buf = (kwy_c *)smem;                 // H+1 complex
  kwy_c *twl = buf + (H + 1);                 // exp(-2 pi i k / H), k < H/8
  double *part = (double *)(twl + TWL);       // MC_FR x 4 x 64 partial sums
  double *cep = part + MC_FR * 256;           // MC_FR x ncut
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * MC_FR;
  for (int i = tid; i < TWL; i += KWY_THREADS) twl[i] = twH[i];
  const kwy_c twb = twN[tid & (N - 1)];
  for (int fr = 0; fr < MC_FR; ++fr) {
    const int64_t frame = f0 + fr;
    if (frame >= T) break;   // uniform
    const double *p = sp + frame * K;
    __syncthreads();
    for (int k = tid; k <= H; k += KWY_THREADS) buf[k] = {log(p[k]), 0.0};
    kwy_irfft_inplace<LOG2N - 1, KWY_THREADS>(buf, twl, twb, twN);
    const double *c = (const double *)buf;  // N * cepstrum (unnormalised c2r)
    for (int n = tid; n < ncut; n += KWY_THREADS) {
      double cn = c[n] / N;
      if (n == 0) cn /= 2.0;
      cep[fr * ncut + n] = cn;
    }
  }
  __syncthreads();
  // mc[j] = sum_n F[n][j] * c[n]; thread (j = tid & 63, part = tid >> 6) strides n by 4
  const int j = tid & 63, q = tid >> 6;
  double acc[MC_FR];
#pragma unroll
  for (int fr = 0; fr < MC_FR; ++fr) acc[fr] = 0.0;
  if (j <= order) {
#pragma unroll 8
    for (int n = q; n < ncut; n += 4) {   // unrolled: eight matrix loads in flight, not one L2 round trip per step
      const double f = F[(size_t)n * MC_STRIDE + j];
#pragma unroll
      for (int fr = 0; fr < MC_FR; ++fr) acc[fr] = __builtin_fma(f, cep[fr * ncut + n], acc[fr]);   // (explicit: the
      // library is built with -ffp-contract=off for the kernels that reproduce serial roundings; this sum is not one)
    }
  }
#pragma unroll
  for (int fr = 0; fr < MC_FR; ++fr) part[fr * 256 + q * 64 + j] = acc[fr];
  __syncthreads();
  for (int e = tid; e < MC_FR * 64; e += KWY_THREADS) {
    const int fr = e >> 6, jj = e & 63;
    const double *pp = part + fr * 256;
    if (jj <= order && f0 + fr < T)
      mc[(f0 + fr) * (order + 1) + jj] = ((pp[jj] + pp[64 + jj]) + pp[128 + jj]) + pp[192 + jj];
  }
}

// ---- any transform length: the dense form ---------------------------------------------------
// The reference resamples features between sampling rates by cutting or padding the spectral axis
// (kwiiyatta/vocoder/abc/synthesizer.py:77-113, vocoder/mcep.py:31-58), so pysptk.sp2mc / mc2sp also see
// spectra of 372, 1024, 3078 ... bins: numpy's irfft / rfft of any even length.  Both conversions are
// linear maps around the log / exp, so for lengths the LDS FFT does not cover the map is
// applied as one dense matrix, built once per (length, order, alpha) on the host in extended precision:
//   sp2mc:  mc[j]    = sum_k G[k][j]  log P[k]        G  = freqt . (c[0] /= 2) . irfft     (K x 64)
//   mc2sp:  log P[k] = sum_j G2[j][k] mc[j]           G2 = rfft.real . mirror . (c[0] *= 2) . freqt
// A frame costs K (order+1) multiply-adds instead of an FFT: still far below its 8 K bytes of traffic.
#define MCD_FR 4
__global__ __launch_bounds__(KWY_THREADS) void k_sp2mc_dense(const double *__restrict__ sp, int64_t T, int K,
                                                            int order, const double *__restrict__ G,
                                                            double *__restrict__ mc) {
  extern __shared__ double smem[];
  double *lg = smem;                       // MCD_FR x K log-spectra
  double *part = lg + (size_t)MCD_FR * K;  // MCD_FR x 4 x 64
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * MCD_FR;
  for (int fr = 0; fr < MCD_FR; ++fr) {
    const int64_t frame = f0 + fr;
    for (int k = tid; k < K; k += KWY_THREADS) lg[(size_t)fr * K + k] = frame < T ? log(sp[frame * K + k]) : 0.0;
  }
  __syncthreads();
  const int j = tid & 63, q = tid >> 6;
  double acc[MCD_FR];
#pragma unroll
  for (int fr = 0; fr < MCD_FR; ++fr) acc[fr] = 0.0;
  if (j <= order) {
#pragma unroll 8
    for (int k = q; k < K; k += 4) {
      const double g = G[(size_t)k * MC_STRIDE + j];
#pragma unroll
      for (int fr = 0; fr < MCD_FR; ++fr) acc[fr] = __builtin_fma(g, lg[(size_t)fr * K + k], acc[fr]);
    }
  }
#pragma unroll
  for (int fr = 0; fr < MCD_FR; ++fr) part[fr * 256 + q * 64 + j] = acc[fr];
  __syncthreads();
  for (int e = tid; e < MCD_FR * 64; e += KWY_THREADS) {
    const int fr = e >> 6, jj = e & 63;
    const double *pp = part + fr * 256;
    if (jj <= order && f0 + fr < T)
      mc[(f0 + fr) * (order + 1) + jj] = ((pp[jj] + pp[64 + jj]) + pp[128 + jj]) + pp[192 + jj];
  }
}

typedef double mc_v4f64 __attribute__((ext_vector_type(4)));
// mc2sp: G2 is [order+1][K].
// The dense map as one f64 MFMA product per 16 frames x 16 bins (v_mfma_f64_16x16x4_f64: A = the frames'
// coefficients, B = G2, order + 1 padded to a multiple of 4 in the k direction), exp applied to the accumulators.
// Used for every length, power of two or not: at K = 1025 the product is 113 MFLOP and G2 (205 KB) stays in L2;
// 15 us for 2201 frames against 35 us of the LDS-FFT form it replaced in round 2.  Lane map: A[row l&15][k l>>4], B[k l>>4][col l&15], D[row (l>>4)+4r][col l&15].
#define MC2_KSTEPS ((MC_MAX_ORDER + 4) / 4)
__global__ __launch_bounds__(KWY_THREADS) void k_mc2sp_mfma(const double *__restrict__ mc, int64_t T, int order,
                                                           int K, const double *__restrict__ G2,
                                                           double *__restrict__ sp) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ar = lane & 15, ak = lane >> 4;
  const int64_t t0 = (int64_t)blockIdx.x * 16;
  const int ksteps = (order + 4) / 4;
  double a[MC2_KSTEPS];
#pragma unroll
  for (int ks = 0; ks < MC2_KSTEPS; ++ks) {
    const int j = 4 * ks + ak;
    a[ks] = (ks < ksteps && j <= order && t0 + ar < T) ? mc[(t0 + ar) * (order + 1) + j] : 0.0;
  }
  const int ntile = (K + 15) / 16;
  for (int bt = blockIdx.y * KWY_WAVES + wv; bt < ntile; bt += gridDim.y * KWY_WAVES) {
    const int b0 = bt * 16;
    mc_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < MC2_KSTEPS; ++ks) {
      if (ks < ksteps) {
        const int j = 4 * ks + ak;
        const double b = (j <= order && b0 + ar < K) ? G2[(size_t)j * K + b0 + ar] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t t = t0 + ak + 4 * r;
      if (t < T && b0 + ar < K) sp[t * K + b0 + ar] = exp(acc[r]);
    }
  }
}

// ---- host side ----------------------------------------------------------------------
int kwy_get_sp2mc_matrix(kwy_ctx *ctx, int N, int order, double alpha, const double **out, int *ncut) {
  char key[96];
  snprintf(key, sizeof(key), "sp2mc:%d:%d:%.17g", N, order, alpha);
  std::string cnt_key = std::string(key) + ":ncut";
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    std::vector<double> F;
    freqt_matrix(N, order, alpha, F);
    // columns beyond ncut contribute < 1e-40 of a unit cepstral value: drop them
    int nc = N;
    while (nc > 1) {
      double mx = 0.0;
      for (int j = 0; j <= order; ++j) mx = fmax(mx, fabs(F[(size_t)(nc - 1) * (order + 1) + j]));
      if (mx > 1e-40) break;
      --nc;
    }
    std::vector<double> Fp((size_t)nc * MC_STRIDE, 0.0);
    for (int n = 0; n < nc; ++n)
      for (int j = 0; j <= order; ++j) Fp[(size_t)n * MC_STRIDE + j] = F[(size_t)n * (order + 1) + j];
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * Fp.size()));
    KWY_HIP(hipMemcpy(d, Fp.data(), sizeof(double) * Fp.size(), hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
    ctx->i_vals[cnt_key] = nc;
  }
  *out = it->second;
  *ncut = (int)ctx->i_vals[cnt_key];
  return KWY_OK;
}

template <int LOG2N>
static int launch_sp2mc(kwy_ctx *ctx, const double *sp, int64_t T, int order, const double *F, int ncut,
                        double *mc) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  const kwy_c *twH, *twN;
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N - 1, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N, &twN));
  size_t lds = sizeof(kwy_c) * ((H + 1) + (H / 8 > 1 ? H / 8 : 1)) + sizeof(double) * (size_t)MC_FR * (256 + ncut);
  if (lds > 160 * 1024) { ctx->err = "sp2mc: transform too long for the LDS"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_sp2mc<LOG2N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_sp2mc", hipLaunchKernelGGL(k_sp2mc<LOG2N>, dim3((unsigned)((T + MC_FR - 1) / MC_FR)), dim3(KWY_THREADS), lds, ctx->stream, sp, T, order, F,
                     ncut, twH, twN, mc));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

// cos(2 pi m / N), m < N, in extended precision (first octant evaluated, the rest by symmetry of the index)
static void cos_table(int N, std::vector<long double> &c) {
  c.resize(N);
  const long double w = 2.0L * 3.141592653589793238462643383279502884L / (long double)N;
  for (int m = 0; m < N; ++m) c[m] = cosl(w * (long double)(m <= N - m ? m : N - m));
}

// G[k][j] (K x MC_STRIDE): mel-cepstral coefficient j per unit of log P[k], any even N = 2 (K - 1)
static std::mutex g_dense_mutex;
static std::map<std::string, std::vector<double>> g_dense_host;     // built once per process and (K, order, alpha):
                                                                    // the batch drivers hold one context per stream
static int get_sp2mc_dense(kwy_ctx *ctx, int K, int order, double alpha, const double **out) {
  char key[96];
  snprintf(key, sizeof(key), "sp2mc_dense:%d:%d:%.17g", K, order, alpha);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    std::lock_guard<std::mutex> guard(g_dense_mutex);
    std::vector<double> &G = g_dense_host[key];
    if (G.empty()) {
      const int N = 2 * (K - 1), H = K - 1;
      std::vector<double> F;
      freqt_matrix(N, order, alpha, F);        // cepstral index n (< N) -> coefficient j
      int nc = N;
      while (nc > 1) {
        double mx = 0.0;
        for (int j = 0; j <= order; ++j) mx = fmax(mx, fabs(F[(size_t)(nc - 1) * (order + 1) + j]));
        if (mx > 1e-40) break;
        --nc;
      }
      std::vector<long double> cs;
      cos_table(N, cs);
      G.assign((size_t)K * MC_STRIDE, 0.0);
      std::vector<long double> acc(order + 1);
      for (int k = 0; k < K; ++k) {
        // irfft: c[n] = (1/N) sum_k w_k L[k] cos(2 pi n k / N), w = 1 at k = 0 and k = H, else 2
        const long double wk = ((k == 0 || k == H) ? 1.0L : 2.0L) / (long double)N;
        for (int j = 0; j <= order; ++j) acc[j] = 0.0L;
        for (int n = 0; n < nc; ++n) {
          const long double cn = wk * cs[(int)(((int64_t)n * k) % N)] * (n == 0 ? 0.5L : 1.0L);
          const double *f = &F[(size_t)n * (order + 1)];
          for (int j = 0; j <= order; ++j) acc[j] += cn * (long double)f[j];
        }
        for (int j = 0; j <= order; ++j) G[(size_t)k * MC_STRIDE + j] = (double)acc[j];
      }
    }
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * G.size()));
    KWY_HIP(hipMemcpy(d, G.data(), sizeof(double) * G.size(), hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  *out = it->second;
  return KWY_OK;
}


static int get_mc2sp_dense(kwy_ctx *ctx, int K, int order, double alpha, const double **out) {
  char key[96];
  snprintf(key, sizeof(key), "mc2sp_dense:%d:%d:%.17g", K, order, alpha);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    std::lock_guard<std::mutex> guard(g_dense_mutex);
    std::vector<double> &G2 = g_dense_host[key];
    if (G2.empty()) {
      const int N = 2 * (K - 1), H = K - 1;
      std::vector<double> F;                   // [order+1][H+1]: coefficient j -> cepstral index i
      freqt_matrix(order + 1, H, -alpha, F);
      std::vector<long double> cs;
      cos_table(N, cs);
      G2.assign((size_t)(order + 1) * K, 0.0);
      std::vector<long double> acc(order + 1);
      for (int k = 0; k < K; ++k) {
        // rfft of the mirrored sequence: S[k] = 2 c0 + 2 sum_{0<i<H} c_i cos(2 pi i k / N) + c_H cos(pi k)
        for (int j = 0; j <= order; ++j) acc[j] = 0.0L;
        for (int i = 0; i <= H; ++i) {
          const long double w = ((i == H) ? 1.0L : 2.0L) * cs[(int)(((int64_t)i * k) % N)];
          for (int j = 0; j <= order; ++j) acc[j] += w * (long double)F[(size_t)j * (H + 1) + i];
        }
        for (int j = 0; j <= order; ++j) G2[(size_t)j * K + k] = (double)acc[j];
      }
    }
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * G2.size()));
    KWY_HIP(hipMemcpy(d, G2.data(), sizeof(double) * G2.size(), hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  *out = it->second;
  return KWY_OK;
}

static int launch_sp2mc_dense(kwy_ctx *ctx, const double *sp, int64_t T, int K, int order, double alpha, double *mc) {
  const double *G;
  KWY_TRY(get_sp2mc_dense(ctx, K, order, alpha, &G));
  const size_t lds = sizeof(double) * ((size_t)MCD_FR * K + MCD_FR * 256);
  KWY_HIP(hipFuncSetAttribute((const void *)k_sp2mc_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_PROF(ctx, "k_sp2mc_dense", hipLaunchKernelGGL(k_sp2mc_dense, dim3((unsigned)((T + MCD_FR - 1) / MCD_FR)), dim3(KWY_THREADS), lds, ctx->stream, sp, T, K, order, G, mc));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int launch_mc2sp_dense(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, int K, double *sp) {
  const double *G2;
  KWY_TRY(get_mc2sp_dense(ctx, K, order, alpha, &G2));
  const unsigned gx = (unsigned)((T + 15) / 16);
  const unsigned gy = gx >= 512 ? 1 : (gx >= 256 ? 2 : 4);        // enough workgroups for the chip at short T too
  KWY_PROF(ctx, "k_mc2sp", hipLaunchKernelGGL(k_mc2sp_mfma, dim3(gx, gy), dim3(KWY_THREADS), 0, ctx->stream, mc, T, order, K, G2, sp));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int mcep_check(kwy_ctx *ctx, const void *a, const void *b, int64_t T, int K, int order, double alpha,
                      int *log2n) {
  if (!ctx) return KWY_EINVAL;
  const int N = 2 * (K - 1);
  if (!a || !b || T <= 0 || K < 2 || K > 4097) {
    ctx->err = "mcep: need 2 <= spectrum length <= 4097";
    return KWY_EINVAL;
  }
  if (order < 1 || order > MC_MAX_ORDER || order > N / 2) { ctx->err = "mcep: order out of range"; return KWY_EINVAL; }
  if (!(fabs(alpha) < 1.0)) { ctx->err = "mcep: |alpha| must be < 1"; return KWY_EINVAL; }
  const int l = kwy_ilog2(N);
  *log2n = ((1 << l) == N && l >= 9 && l <= 13) ? l : 0;   // 0: no LDS FFT of that length -> dense form
  return KWY_OK;
}

extern "C" int kwy_sp2mc_dev(kwy_ctx *ctx, const double *sp, int64_t T, int K, int order, double alpha,
                             double *mc) {
  int l;
  KWY_TRY(mcep_check(ctx, sp, mc, T, K, order, alpha, &l));
  KWY_HIP(hipSetDevice(ctx->device));
  if (l == 0) return launch_sp2mc_dense(ctx, sp, T, K, order, alpha, mc);
  const double *F;
  int ncut;
  KWY_TRY(kwy_get_sp2mc_matrix(ctx, 1 << l, order, alpha, &F, &ncut));
  switch (l) {
    case 9: return launch_sp2mc<9>(ctx, sp, T, order, F, ncut, mc);
    case 10: return launch_sp2mc<10>(ctx, sp, T, order, F, ncut, mc);
    case 11: return launch_sp2mc<11>(ctx, sp, T, order, F, ncut, mc);
    case 12: return launch_sp2mc<12>(ctx, sp, T, order, F, ncut, mc);
    default: return launch_sp2mc<13>(ctx, sp, T, order, F, ncut, mc);
  }
}

extern "C" int kwy_mc2sp_dev(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, int fftlen,
                             double *sp) {
  int l;
  KWY_TRY(mcep_check(ctx, mc, sp, T, fftlen / 2 + 1, order, alpha, &l));
  KWY_HIP(hipSetDevice(ctx->device));
  (void)l;      // one form for every length: the dense map on the matrix cores
  return launch_mc2sp_dense(ctx, mc, T, order, alpha, fftlen / 2 + 1, sp);
}

extern "C" int kwy_sp2mc(kwy_ctx *ctx, const double *sp, int64_t T, int K, int order, double alpha,
                         double *mc) {
  int l;
  KWY_TRY(mcep_check(ctx, sp, mc, T, K, order, alpha, &l));
  KWY_HIP(hipSetDevice(ctx->device));
  size_t bs = kwy_pad(sizeof(double) * T * K), bm = kwy_pad(sizeof(double) * T * (order + 1));
  KWY_TRY(kwy_arena_begin(ctx, bs + bm));
  double *dsp = kwy_arena<double>(ctx, (size_t)T * K), *dmc = kwy_arena<double>(ctx, (size_t)T * (order + 1));
  KWY_HIP(hipMemcpyAsync(dsp, sp, sizeof(double) * T * K, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_sp2mc_dev(ctx, dsp, T, K, order, alpha, dmc));
  KWY_HIP(hipMemcpyAsync(mc, dmc, sizeof(double) * T * (order + 1), hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}

extern "C" int kwy_mc2sp(kwy_ctx *ctx, const double *mc, int64_t T, int order, double alpha, int fftlen,
                         double *sp) {
  int l;
  const int K = fftlen / 2 + 1;
  KWY_TRY(mcep_check(ctx, mc, sp, T, K, order, alpha, &l));
  KWY_HIP(hipSetDevice(ctx->device));
  size_t bs = kwy_pad(sizeof(double) * T * K), bm = kwy_pad(sizeof(double) * T * (order + 1));
  KWY_TRY(kwy_arena_begin(ctx, bs + bm));
  double *dsp = kwy_arena<double>(ctx, (size_t)T * K), *dmc = kwy_arena<double>(ctx, (size_t)T * (order + 1));
  KWY_HIP(hipMemcpyAsync(dmc, mc, sizeof(double) * T * (order + 1), hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_mc2sp_dev(ctx, dmc, T, order, alpha, fftlen, dsp));
  KWY_HIP(hipMemcpyAsync(sp, dsp, sizeof(double) * T * K, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
