// kwy_align.hip -- the small device-side pieces around FastDTW that let a whole
// source/target pair stay resident in HBM between analysis and synthesis:
//
//   kwy_align_features_dev  DTW feature rows [power term, voicing term, mc1..mcN]
//                           (reference: make_feature/binalize, kwiiyatta/vocoder/align.py:10-58,
//                            power='binalize', power_pivot='max', vuv='f0')
//   kwy_align_project_dev   DTW path -> one source index per target frame
//                           (reference: project_path_iter, kwiiyatta/vocoder/align.py:99-120)
//   kwy_gather_rows_dev     row gather of a (T, width) matrix
//                           (reference: Feature.__getitem__, kwiiyatta/vocoder/abc/feature.py:170-194)
//
// All three are index/byte work: HBM-bound row copies and one short serial walk.
#include "kwy_internal.hpp"

__global__ __launch_bounds__(KWY_THREADS) void k_align_features(const double *__restrict__ mc, int64_t T,
                                                               int ncoef, const double *__restrict__ f0,
                                                               double power_weight, double power_threshold,
                                                               double vuv_weight, double *__restrict__ out) {
  __shared__ double red[KWY_WAVES];
  __shared__ double s_thr;
  const int tid = threadIdx.x;
  double mx = -INFINITY;
  for (int64_t t = tid; t < T; t += KWY_THREADS) mx = fmax(mx, mc[t * ncoef]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    double m = red[0];
    for (int i = 1; i < KWY_WAVES; ++i) m = fmax(m, red[i]);
    s_thr = m - power_threshold;
  }
  __syncthreads();
  const double thr = s_thr;
  const int w = ncoef + 1;
  for (int64_t e = tid; e < T * w; e += KWY_THREADS) {
    const int64_t t = e / w;
    const int c = (int)(e % w);
    double v;
    if (c == 0) v = mc[t * ncoef] >= thr ? power_weight : 0.0;
    else if (c == 1) v = f0[t] > 0 ? vuv_weight : 0.0;
    else v = mc[t * ncoef + (c - 1)];
    out[e] = v;
  }
}

// serial walk over the path (one thread): exactly project_path_iter
__global__ void k_align_project(const int32_t *__restrict__ path, const int64_t *__restrict__ path_len,
                                int trim_len, int32_t *__restrict__ idx, int64_t cap,
                                int64_t *__restrict__ n_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int64_t L = *path_len;
  int64_t n = 0;
  if (L > 0) {
    long long prev_x = -1, prev_y = -1;
    long long len_y = path[2 * (L - 1) + 1] + 1;
    if (trim_len > 0) { prev_y += trim_len; len_y -= trim_len; }
    for (int64_t k = 0; k < L; ++k) {
      long long x = path[2 * k], y = path[2 * k + 1];
      if (y <= prev_y) continue;
      if (y - prev_y > 1) {
        if (y > len_y - 1) y = len_y - 1;
        const long long diff_x = x - prev_x, diff_y = y - prev_y;
        for (long long i = 0; i < diff_y; ++i) {
          // Python floor division; diff_y - 1 >= 1 here unless the clamp made diff_y <= 1
          long long den = diff_y - 1;
          long long num = diff_x * i;
          long long q = den != 0 ? (num >= 0 ? num / den : -((-num + den - 1) / den)) : 0;
          if (n < cap) idx[n] = (int32_t)(prev_x + q);
          ++n;
        }
      } else if (y >= len_y) {
        break;
      } else {
        if (n < cap) idx[n] = (int32_t)x;
        ++n;
      }
      prev_x = x;
      prev_y = y;
    }
  }
  *n_out = n;
}

__global__ void k_gather_rows(const double *__restrict__ src, int64_t src_rows, int width,
                              const int32_t *__restrict__ idx, int64_t n, double *__restrict__ dst) {
  const int64_t row = blockIdx.x;
  if (row >= n) return;
  int64_t r = idx[row];
  if (r < 0) r = 0;
  if (r >= src_rows) r = src_rows - 1;
  const double *s = src + r * width;
  double *d = dst + row * width;
  for (int c = threadIdx.x; c < width; c += blockDim.x) d[c] = s[c];
}

extern "C" int kwy_align_features_dev(kwy_ctx *ctx, const double *mc, int64_t T, int ncoef, const double *f0,
                                      double power_weight, double power_threshold, double vuv_weight,
                                      double *out) {
  if (!ctx) return KWY_EINVAL;
  if (!mc || !f0 || !out || T <= 0 || ncoef < 1) { ctx->err = "align_features: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_align_features, dim3(1), dim3(KWY_THREADS), 0, ctx->stream, mc, T, ncoef, f0,
                     power_weight, power_threshold, vuv_weight, out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_align_project_dev(kwy_ctx *ctx, const int32_t *path, const int64_t *path_len, int trim_len,
                                     int32_t *idx, int64_t idx_capacity, int64_t *n_out) {
  if (!ctx) return KWY_EINVAL;
  if (!path || !path_len || !idx || !n_out || trim_len < 0 || idx_capacity <= 0) {
    ctx->err = "align_project: bad argument";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_align_project, dim3(1), dim3(64), 0, ctx->stream, path, path_len, trim_len, idx,
                     idx_capacity, n_out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_gather_rows_dev(kwy_ctx *ctx, const double *src, int64_t src_rows, int width,
                                   const int32_t *idx, int64_t n, double *dst) {
  if (!ctx) return KWY_EINVAL;
  if (!src || !idx || !dst || src_rows <= 0 || width <= 0 || n < 0) { ctx->err = "gather_rows: bad argument"; return KWY_EINVAL; }
  if (n == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n), dim3(width >= 256 ? 256 : 64), 0, ctx->stream, src,
                     src_rows, width, idx, n, dst);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}
