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
// All three are index/byte work: HBM-bound row copies and a pass over the path.
#include "kwy_internal.hpp"

// Every workgroup finds the power threshold (max over all frames of c0, minus the offset) for itself -- T loads of
// an L2-resident column -- and then writes the feature rows of its 8 frames: one launch instead of a reduction
// kernel followed by a row kernel.
__device__ __forceinline__ void align_features_body(const double *__restrict__ mc, int64_t T,
                                                    int ncoef, const double *__restrict__ f0,
                                                    double power_threshold, double power_weight,
                                                    double vuv_weight, double *__restrict__ out) {
  __shared__ double red[KWY_WAVES];
  const int tid = threadIdx.x;
  double mx = -INFINITY;
  for (int64_t t = tid; t < T; t += KWY_THREADS) mx = fmax(mx, mc[t * ncoef]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  double m = red[0];
  for (int i = 1; i < KWY_WAVES; ++i) m = fmax(m, red[i]);
  const double thr = m - power_threshold;
  const int64_t t = (int64_t)blockIdx.x * 8 + (tid >> 5);  // 8 frames per workgroup, 32 lanes per frame
  if (t >= T) return;
  const int w = ncoef + 1;
  for (int c = tid & 31; c < w; c += 32) {
    double v;
    if (c == 0) v = mc[t * ncoef] >= thr ? power_weight : 0.0;
    else if (c == 1) v = f0[t] > 0 ? vuv_weight : 0.0;
    else v = mc[t * ncoef + (c - 1)];
    out[t * w + c] = v;
  }
}

// project_path_iter.  A DTW path moves by at most one row and one column per cell: then the generator
// yields, for every target frame y in [trim, last_y - trim], the x of the first path cell with that y
// -- independent per cell, done by all threads.  Any other path (a jump in y, a y that goes back) takes
// the reference's loop literally: the first wavefront stages path tiles in LDS, lane 0 walks them, and
// the produced indices leave through LDS as well.
#define AL_TILE 1024
#define AL_NT 256
__device__ __forceinline__ void align_project_body(const int32_t *__restrict__ path,
                                                   const int64_t *__restrict__ path_len, int trim_len,
                                                   int32_t *__restrict__ idx, int64_t cap,
                                                   int64_t *__restrict__ n_out) {
  __shared__ int32_t sp[2 * AL_TILE];
  __shared__ int32_t so[AL_TILE];
  __shared__ long long st[4];  // prev_x, prev_y, n, stop
  __shared__ int s_irregular;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t L = *path_len;
  if (L <= 0) { if (tid == 0) *n_out = 0; return; }
  long long len_y = path[2 * (L - 1) + 1] + 1;
  const long long prev_y0 = -1 + (trim_len > 0 ? trim_len : 0);
  if (trim_len > 0) len_y -= trim_len;
  if (tid == 0) s_irregular = 0;
  __syncthreads();
  {
    bool bad = false;
    for (int64_t k = tid; k < L; k += AL_NT) {
      const long long y = path[2 * k + 1], yp = k > 0 ? path[2 * k - 1] : -1;
      if (y < yp || y - yp > 1) bad = true;
    }
    if (bad) s_irregular = 1;
  }
  __syncthreads();
  if (!s_irregular) {
    for (int64_t k = tid; k < L; k += AL_NT) {
      const long long y = path[2 * k + 1], yp = k > 0 ? path[2 * k - 1] : -1;
      if (y > yp && y > prev_y0 && y < len_y) {
        const long long n = y - (prev_y0 + 1);
        if (n < cap) idx[n] = path[2 * k];
      }
    }
    if (tid == 0) { const long long n = len_y - (prev_y0 + 1); *n_out = n > 0 ? n : 0; }
    return;
  }
  if (tid >= 64) return;   // the general case: one wavefront, no workgroup barrier below
  if (lane == 0) { st[0] = -1; st[1] = prev_y0; st[2] = 0; st[3] = 0; }
  int64_t k0 = 0;
  while (k0 < L) {
    const int nk = (int)min((int64_t)AL_TILE, L - k0);
    for (int e = lane; e < 2 * nk; e += 64) sp[e] = path[2 * k0 + e];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // lane 0 consumes path cells until the tile is exhausted or the output tile is full
    __shared__ int s_used, s_made;
    if (lane == 0) {
      long long prev_x = st[0], prev_y = st[1];
      int made = 0, k = 0;
      bool stop = false;
      for (; k < nk && !stop; ++k) {
        long long x = sp[2 * k], y = sp[2 * k + 1];
        if (y <= prev_y) continue;
        if (y - prev_y > 1) {
          if (y > len_y - 1) y = len_y - 1;
          const long long diff_x = x - prev_x, diff_y = y - prev_y;
          if (made + diff_y > AL_TILE) {
            if (made > 0) break;         // flush what we have, redo this cell next round
            // a single gap longer than the tile: emit directly
          }
          for (long long i = 0; i < diff_y; ++i) {
            const long long den = diff_y - 1, num = diff_x * i;
            const long long q = den != 0 ? (num >= 0 ? num / den : -((-num + den - 1) / den)) : 0;
            if (made < AL_TILE) so[made++] = (int32_t)(prev_x + q);
          }
        } else if (y >= len_y) {
          stop = true;
          break;
        } else {
          if (made >= AL_TILE) break;
          so[made++] = (int32_t)x;
        }
        prev_x = x;
        prev_y = y;
      }
      st[0] = prev_x; st[1] = prev_y;
      s_used = k; s_made = made;
      if (stop) st[3] = 1;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int made = s_made, used = s_used;
    const long long n0 = st[2];
    for (int e = lane; e < made; e += 64)
      if (n0 + e < cap) idx[n0 + e] = so[e];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) st[2] = n0 + made;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (st[3]) break;
    k0 += used > 0 ? used : 1;
  }
  if (lane == 0) *n_out = st[2];
}

// The three kernels take their jobs by value (up to AL_BATCH per launch, blockIdx.y = job): the pairs of a batch
// driver share one launch, a single call is a batch of one.
#define AL_BATCH 64
template <class JOB>
struct al_jobs { int n; JOB j[AL_BATCH]; };

__global__ __launch_bounds__(KWY_THREADS) void k_align_features(al_jobs<kwy_align_job> b, int ncoef,
                                                               double power_threshold, double power_weight,
                                                               double vuv_weight) {
  const kwy_align_job &q = b.j[blockIdx.y];
  if ((int64_t)blockIdx.x * 8 >= q.T) return;
  align_features_body(q.mc, q.T, ncoef, q.f0, power_threshold, power_weight, vuv_weight, q.out);
}

__global__ __launch_bounds__(AL_NT) void k_align_project(al_jobs<kwy_project_job> b, int trim_len) {
  const kwy_project_job &q = b.j[blockIdx.x];
  align_project_body(q.path, q.path_len, trim_len, q.idx, q.idx_capacity, q.n_out);
}

__global__ void k_gather_rows(al_jobs<kwy_gather_job> b, int width) {
  const kwy_gather_job &q = b.j[blockIdx.y];
  const double *__restrict__ src = q.src;
  const int32_t *__restrict__ idx = q.idx;
  double *__restrict__ dst = q.dst;
  const int64_t src_rows = q.src_rows, n = q.n;
  const int64_t row = blockIdx.x;
  if (row >= n) return;
  int64_t r = idx[row];
  if (r < 0) r = 0;
  if (r >= src_rows) r = src_rows - 1;
  const double *s = src + r * width;
  double *d = dst + row * width;
  for (int c = threadIdx.x; c < width; c += blockDim.x) d[c] = s[c];
}

template <class JOB, class LAUNCH>
static int al_for_batches(const JOB *jobs, int count, LAUNCH launch) {
  for (int j0 = 0; j0 < count; j0 += AL_BATCH) {
    al_jobs<JOB> b;
    b.n = count - j0 < AL_BATCH ? count - j0 : AL_BATCH;
    for (int k = 0; k < AL_BATCH; ++k) b.j[k] = jobs[j0 + (k < b.n ? k : 0)];
    launch(b);
  }
  return KWY_OK;
}

extern "C" int kwy_align_features_batch_dev(kwy_ctx *ctx, const kwy_align_job *jobs, int count, int ncoef,
                                            double power_weight, double power_threshold, double vuv_weight) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || ncoef < 1) { ctx->err = "align_features: bad argument"; return KWY_EINVAL; }
  for (int j = 0; j < count; ++j)
    if (!jobs[j].mc || !jobs[j].f0 || !jobs[j].out || jobs[j].T <= 0) { ctx->err = "align_features: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  al_for_batches(jobs, count, [&](const al_jobs<kwy_align_job> &b) {
    int64_t T = 0;
    for (int k = 0; k < b.n; ++k) T = b.j[k].T > T ? b.j[k].T : T;
    hipLaunchKernelGGL(k_align_features, dim3((unsigned)((T + 7) / 8), b.n), dim3(KWY_THREADS), 0, ctx->stream, b, ncoef,
                       power_threshold, power_weight, vuv_weight);
  });
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_align_features_dev(kwy_ctx *ctx, const double *mc, int64_t T, int ncoef, const double *f0,
                                      double power_weight, double power_threshold, double vuv_weight,
                                      double *out) {
  const kwy_align_job job = {mc, f0, T, out};
  return kwy_align_features_batch_dev(ctx, &job, 1, ncoef, power_weight, power_threshold, vuv_weight);
}

extern "C" int kwy_align_project_batch_dev(kwy_ctx *ctx, const kwy_project_job *jobs, int count, int trim_len) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || trim_len < 0) { ctx->err = "align_project: bad argument"; return KWY_EINVAL; }
  for (int j = 0; j < count; ++j)
    if (!jobs[j].path || !jobs[j].path_len || !jobs[j].idx || !jobs[j].n_out || jobs[j].idx_capacity <= 0) {
      ctx->err = "align_project: bad argument";
      return KWY_EINVAL;
    }
  KWY_HIP(hipSetDevice(ctx->device));
  al_for_batches(jobs, count, [&](const al_jobs<kwy_project_job> &b) {
    hipLaunchKernelGGL(k_align_project, dim3(b.n), dim3(AL_NT), 0, ctx->stream, b, trim_len);
  });
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_align_project_dev(kwy_ctx *ctx, const int32_t *path, const int64_t *path_len, int trim_len,
                                     int32_t *idx, int64_t idx_capacity, int64_t *n_out) {
  const kwy_project_job job = {path, path_len, idx, idx_capacity, n_out};
  return kwy_align_project_batch_dev(ctx, &job, 1, trim_len);
}

extern "C" int kwy_gather_rows_batch_dev(kwy_ctx *ctx, const kwy_gather_job *jobs, int count, int width) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || width <= 0) { ctx->err = "gather_rows: bad argument"; return KWY_EINVAL; }
  for (int j = 0; j < count; ++j)
    if (!jobs[j].src || !jobs[j].idx || !jobs[j].dst || jobs[j].src_rows <= 0 || jobs[j].n < 0) {
      ctx->err = "gather_rows: bad argument";
      return KWY_EINVAL;
    }
  KWY_HIP(hipSetDevice(ctx->device));
  al_for_batches(jobs, count, [&](const al_jobs<kwy_gather_job> &b) {
    int64_t n = 0;
    for (int k = 0; k < b.n; ++k) n = b.j[k].n > n ? b.j[k].n : n;
    if (n > 0)
      hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n, b.n), dim3(width >= 256 ? 256 : 64), 0, ctx->stream, b, width);
  });
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_gather_rows_dev(kwy_ctx *ctx, const double *src, int64_t src_rows, int width,
                                   const int32_t *idx, int64_t n, double *dst) {
  const kwy_gather_job job = {src, src_rows, idx, n, dst};
  return kwy_gather_rows_batch_dev(ctx, &job, 1, width);
}
