// kwy_codec.hip -- WORLD aperiodicity band codec on gfx950.
//
// Replaces pyworld.code_aperiodicity / pyworld.decode_aperiodicity, which the reference uses
// only to move aperiodicity between sampling rates or spectrum lengths
// (kwiiyatta/vocoder/world.py:98-145).  WORLD codec.cpp as shipped with pyworld 0.2.8:
//   code   : 20 log10(ap) sampled at 3 kHz, 6 kHz, ... by interp1Q on the bin grid
//   decode : linear interpolation over {0: -60 dB, 3k(i+1): coded[i], fs/2: -1e-12 dB}, 10^(dB/20);
//            frames whose mean coded value exceeds -0.5 dB stay at 1 - 1e-12
// One workgroup per frame; pure streaming (K*8 B in, bands*8 B out and vice versa).
#include <math.h>

#include "kwy_internal.hpp"

#define CODEC_SAFE 0.000000000001
#define CODEC_INTERVAL 3000.0
#define CODEC_UPPER 15000.0
#define CODEC_MAX_BANDS 8

static int codec_num_bands(int fs) {
  double lim = fs / 2.0 - CODEC_INTERVAL;
  if (lim > CODEC_UPPER) lim = CODEC_UPPER;
  int n = (int)(lim / CODEC_INTERVAL);
  return n < 0 ? 0 : n;
}

extern "C" int kwy_aperiodicity_bands(int fs) { return codec_num_bands(fs); }

__global__ __launch_bounds__(64) void k_code_aperiodicity(const double *__restrict__ ap, int64_t T, int fs,
                                                         int fft_size, int nb, double *__restrict__ coded) {
  const int64_t frame = blockIdx.x;
  const int b = threadIdx.x;
  if (b >= nb) return;
  const int K = fft_size / 2 + 1;
  const double *row = ap + frame * K;
  // interp1Q(0, fs/fft, 20 log10(ap), K, 3000 (b+1))
  const double shift = (double)fs / fft_size;
  const double xi = CODEC_INTERVAL * (b + 1.0);
  const int base = (int)((xi - 0) / shift);
  const double frac = (xi - 0) / shift - base;
  const double y0 = 20 * log10(row[base]);
  const double dy = (base >= K - 1) ? 0.0 : 20 * log10(row[base + 1]) - y0;
  coded[frame * nb + b] = y0 + dy * frac;
}

__global__ __launch_bounds__(KWY_THREADS) void k_decode_aperiodicity(const double *__restrict__ coded, int64_t T,
                                                                     int fs, int fft_size, int nb,
                                                                     double *__restrict__ ap) {
  __shared__ double coarse[CODEC_MAX_BANDS + 2];
  __shared__ int s_unvoiced;
  const int64_t frame = blockIdx.x;
  const int tid = threadIdx.x, K = fft_size / 2 + 1;
  double *o = ap + frame * K;
  if (tid == 0) {
    double tmp = 0.0;
    for (int i = 0; i < nb; ++i) { tmp += coded[frame * nb + i]; coarse[i + 1] = coded[frame * nb + i]; }
    coarse[0] = -60.0;
    coarse[nb + 1] = -CODEC_SAFE;
    s_unvoiced = nb <= 0 || (tmp / nb > -0.5);
  }
  __syncthreads();
  if (s_unvoiced) {
    for (int k = tid; k < K; k += KWY_THREADS) o[k] = 1.0 - CODEC_SAFE;
    return;
  }
  const int nn = nb + 2;
  for (int k = tid; k < K; k += KWY_THREADS) {
    const double xi = (double)fs / fft_size * k;
    int seg = 0;  // number of nodes <= xi  (WORLD histc)
    for (int j = 0; j < nn; ++j) {
      const double xj = (j <= nb) ? j * CODEC_INTERVAL : fs / 2.0;
      if (xj <= xi) seg = j + 1;
    }
    if (seg < 1) seg = 1;
    if (seg > nn - 1) seg = nn - 1;
    const double xa = (seg - 1 <= nb) ? (seg - 1) * CODEC_INTERVAL : fs / 2.0;
    const double xb = (seg <= nb) ? seg * CODEC_INTERVAL : fs / 2.0;
    const double s = (xi - xa) / (xb - xa);
    const double v = coarse[seg - 1] + s * (coarse[seg] - coarse[seg - 1]);
    o[k] = pow(10.0, v / 20.0);
  }
}

static int codec_check(kwy_ctx *ctx, const void *a, const void *b, int64_t T, int fs, int fft_size, int nb) {
  if (!ctx) return KWY_EINVAL;
  if (!a || !b || T <= 0 || fs <= 0 || fft_size < 4 || (fft_size & 1) || nb < 0 || nb > CODEC_MAX_BANDS) {
    ctx->err = "aperiodicity codec: bad argument";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_code_aperiodicity_dev(kwy_ctx *ctx, const double *ap, int64_t T, int fs, int fft_size,
                                         double *coded) {
  const int nb = codec_num_bands(fs);
  KWY_TRY(codec_check(ctx, ap, coded, T, fs, fft_size, nb));
  KWY_HIP(hipSetDevice(ctx->device));
  if (nb == 0) return KWY_OK;
  hipLaunchKernelGGL(k_code_aperiodicity, dim3((unsigned)T), dim3(64), 0, ctx->stream, ap, T, fs, fft_size, nb, coded);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_decode_aperiodicity_dev(kwy_ctx *ctx, const double *coded, int64_t T, int fs, int fft_size,
                                           int nb, double *ap) {
  KWY_TRY(codec_check(ctx, coded, ap, T, fs, fft_size, nb));
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_decode_aperiodicity, dim3((unsigned)T), dim3(KWY_THREADS), 0, ctx->stream, coded, T, fs,
                     fft_size, nb, ap);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_code_aperiodicity(kwy_ctx *ctx, const double *ap, int64_t T, int fs, int fft_size,
                                     double *coded) {
  const int nb = codec_num_bands(fs);
  KWY_TRY(codec_check(ctx, ap, coded, T, fs, fft_size, nb));
  KWY_HIP(hipSetDevice(ctx->device));
  if (nb == 0) return KWY_OK;
  const int K = fft_size / 2 + 1;
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * T * K) + kwy_pad(sizeof(double) * T * nb)));
  double *dap = kwy_arena<double>(ctx, (size_t)T * K), *dc = kwy_arena<double>(ctx, (size_t)T * nb);
  KWY_HIP(hipMemcpyAsync(dap, ap, sizeof(double) * T * K, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_code_aperiodicity_dev(ctx, dap, T, fs, fft_size, dc));
  KWY_HIP(hipMemcpyAsync(coded, dc, sizeof(double) * T * nb, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}

extern "C" int kwy_decode_aperiodicity(kwy_ctx *ctx, const double *coded, int64_t T, int fs, int fft_size, int nb,
                                       double *ap) {
  KWY_TRY(codec_check(ctx, coded, ap, T, fs, fft_size, nb));
  KWY_HIP(hipSetDevice(ctx->device));
  const int K = fft_size / 2 + 1;
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(double) * T * K) + kwy_pad(sizeof(double) * T * (nb + 1))));
  double *dap = kwy_arena<double>(ctx, (size_t)T * K), *dc = kwy_arena<double>(ctx, (size_t)T * (nb + 1));
  if (nb > 0) KWY_HIP(hipMemcpyAsync(dc, coded, sizeof(double) * T * nb, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_decode_aperiodicity_dev(ctx, dc, T, fs, fft_size, nb, dap));
  KWY_HIP(hipMemcpyAsync(ap, dap, sizeof(double) * T * K, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
