// kwy_train.hip -- device-side pieces of the TRAINING-SET path, so that a parallel corpus goes from waveforms
// to the joint feature matrix of the converter fit without leaving HBM:
//
//   kwy_trim_length_dev     TrimmedDataset: number of frames kept after nnmnkwii's trim_zeros_frames on the
//                           spectral envelope            (kwiiyatta/converter/dataset.py:49-52)
//   kwy_is_voiced_dev       WorldSynthesizer.extract_is_voiced   (kwiiyatta/vocoder/world.py:147-151)
//   kwy_align_even_dev      dtw_feature's `strict` filtering of the FastDTW path + align_even's cut to the
//                           un-padded stretch: two index lists    (kwiiyatta/vocoder/align.py:73-92, 134-146)
//   kwy_delta_features_dev  nnmnkwii delta_features with the reference's DELTA_WINDOWS
//                                                                 (kwiiyatta/converter/delta.py:8-12, 30)
//   kwy_joint_rows_dev      np.hstack((x, y)) + remove_zeros_frames, appended in order
//                                                                 (kwiiyatta/converter/dataset.py:61-77)
//
// Index and byte work on a few thousand frames per utterance: single-workgroup scans and row copies.
#include <math.h>

#include <algorithm>

#include "kwy_internal.hpp"

#define TR_NT 256

// exclusive scan of one int per thread over the workgroup; *total = sum.  sh: TR_NT/64 + 1 ints
__device__ __forceinline__ int tr_block_exscan(int v, int *sh, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = (int)kwy_wave_scan_u32((uint32_t)v);
  __syncthreads();
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TR_NT / 64; ++w) {
    if (w < wv) base += sh[w];
    tot += sh[w];
  }
  *total = tot;
  return base + inc - v;
}

// rowsum[t] = sum_k |sp[t][k]|   (one wavefront per row)
__device__ __forceinline__ void tr_rowsum_body(const double *__restrict__ sp, int64_t T, int K,
                                               double *__restrict__ rowsum) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wv;
  if (t >= T) return;
  double s = 0.0;
  for (int k = lane; k < K; k += 64) s += fabs(sp[t * K + k]);
  s = kwy_wave_sum(s);
  if (lane == 0) rowsum[t] = s;
}

// len(np.trim_zeros(s)) with s[s < eps] = 0: T minus the leading and the trailing run of "zero" rows
__device__ __forceinline__ void tr_trim_body(const double *__restrict__ rowsum, int64_t T, double eps,
                                             int64_t *__restrict__ n_out) {
  __shared__ long long first, last;
  if (threadIdx.x == 0) { first = T; last = -1; }
  __syncthreads();
  long long f = T, l = -1;
  for (int64_t t = threadIdx.x; t < T; t += TR_NT)
    if (!(rowsum[t] < eps)) { if (t < f) f = t; if (t > l) l = t; }
  if (f < T) atomicMin(&first, f);
  if (l >= 0) atomicMax(&last, l);
  __syncthreads();
  if (threadIdx.x == 0) n_out[0] = last >= first ? last - first + 1 : 0;
}

__global__ void k_tr_is_voiced(const double *__restrict__ f0, const double *__restrict__ ap, int64_t T, int K,
                               double lowest_f0, double *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  out[t] = (f0[t] >= lowest_f0 && ap[t * K] <= 0.999) ? 1.0 : 0.0;
}

// The reference's `check` (align.py:73-80, with its y_feature[y, 0] in the voicing test kept as written) on the
// inner cells of the path, then align_even's [begin, end) cut; the survivors' x and y go to idx_x / idx_y in
// order.  One workgroup; three passes over the path (positions, then the two boundaries, then the copy).
__device__ __forceinline__ void tr_align_even_body(const int32_t *__restrict__ path,
                                                   const int64_t *__restrict__ path_len,
                                                   const double *__restrict__ fx, const double *__restrict__ fy,
                                                   int width, int strict, int use_power, int use_vuv,
                                                   int64_t Tx, int64_t Ty, int pad_len, int32_t *__restrict__ pos,
                                                   int32_t *__restrict__ idx_x, int32_t *__restrict__ idx_y,
                                                   int64_t cap, int64_t *__restrict__ n_out) {
  __shared__ int sh[TR_NT / 64 + 1];
  __shared__ long long s_begin, s_end;
  __shared__ int s_run;
  const int tid = threadIdx.x;
  const int64_t L = path_len[0];
  if (L <= 0) { if (tid == 0) n_out[0] = 0; return; }
  // np.fromiter(chain(path[0], <kept inner cells>, path[-1])): a one-cell path yields that cell twice
  const int64_t Lv = L == 1 ? 2 : L;            // virtual length: cell v = path[min(v, L - 1)]
  auto keep = [&](int64_t v) -> bool {
    if (!strict || v == 0 || v == Lv - 1) return true;
    const int x = path[2 * v], y = path[2 * v + 1];
    if (use_power && ((fx[(size_t)x * width] > 0) != (fy[(size_t)y * width] > 0))) return false;
    if (use_vuv && ((fx[(size_t)x * width + 1] > 0) != (fy[(size_t)y * width] > 0))) return false;
    return true;
  };
  // pass 1: position of every kept cell in the filtered path (pos[v] = -1 for dropped cells)
  if (tid == 0) { s_run = 0; s_begin = -1; s_end = -1; }
  __syncthreads();
  for (int64_t v0 = 0; v0 < Lv; v0 += TR_NT) {
    const int64_t v = v0 + tid;
    const int k = (v < Lv && keep(v)) ? 1 : 0;
    int tot;
    const int off = tr_block_exscan(k, sh, &tot);
    const int run = s_run;
    if (v < Lv) pos[v] = k ? run + off : -1;
    __syncthreads();
    if (tid == 0) s_run = run + tot;
    __syncthreads();
  }
  const long long nf = s_run;
  // pass 2: begin = first filtered position with x >= pad and y >= pad; end = first with x >= Tx - pad and
  // y >= Ty - pad (np.argmax of a boolean array: 0 when nothing is true)
  long long b = 1ll << 40, e = 1ll << 40;
  for (int64_t v = tid; v < Lv; v += TR_NT) {
    const int p = pos[v];
    if (p < 0) continue;
    const int64_t c = v < L ? v : L - 1;
    const int x = path[2 * c], y = path[2 * c + 1];
    if (x >= pad_len && y >= pad_len && p < b) b = p;
    if (x >= Tx - pad_len && y >= Ty - pad_len && p < e) e = p;
  }
  if (tid == 0) { s_begin = 1ll << 40; s_end = 1ll << 40; }
  __syncthreads();
  if (b < (1ll << 40)) atomicMin(&s_begin, b);
  if (e < (1ll << 40)) atomicMin(&s_end, e);
  __syncthreads();
  long long begin = s_begin < (1ll << 40) ? s_begin : 0, end = s_end < (1ll << 40) ? s_end : 0;
  if (pad_len <= 0) { begin = 0; end = nf; }
  // pass 3
  for (int64_t v = tid; v < Lv; v += TR_NT) {
    const int p = pos[v];
    if (p < begin || p >= end) continue;
    const int64_t c = v < L ? v : L - 1;
    const long long o = p - begin;
    if (o < cap) { idx_x[o] = path[2 * c]; idx_y[o] = path[2 * c + 1]; }
  }
  if (tid == 0) { const long long n = end > begin ? end - begin : 0; n_out[0] = n < cap ? n : cap; }
}

// out[t] = [x[t], -0.5 x[t-1] + 0.5 x[t+1], x[t-1] - 2 x[t] + x[t+1]] with zeros outside [0, n)
// (np.correlate(x, w, 'same') per column: the products are summed in tap order)
__global__ void k_tr_delta(const double *__restrict__ x, const int64_t *__restrict__ n_p, int64_t cap, int d,
                           double *__restrict__ out) {
  const int64_t n = min(n_p[0], cap);
  const int64_t t = blockIdx.x;
  if (t >= n) return;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const double xm = t > 0 ? x[(t - 1) * d + c] : 0.0, x0 = x[t * d + c], xp = t + 1 < n ? x[(t + 1) * d + c] : 0.0;
    double *o = out + t * 3 * d;
    o[c] = x0;
    o[d + c] = (-0.5 * xm + 0.0 * x0) + 0.5 * xp;
    o[2 * d + c] = (1.0 * xm + -2.0 * x0) + 1.0 * xp;
  }
}

// joint[r] = [xd[t] | yd[t]] for the rows t < n whose L1 norm is >= eps, in order; n_out = number of rows
template <bool WRITE>
__device__ __forceinline__ void tr_joint_body(const double *__restrict__ xd, const double *__restrict__ yd,
                                              const int64_t *__restrict__ n_p, int64_t cap, int w, double eps,
                                              double *__restrict__ joint, int64_t *__restrict__ n_out) {
  __shared__ int sh[TR_NT / 64 + 1];
  __shared__ int s_run;
  __shared__ int s_dst[TR_NT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t n = min(n_p[0], cap);
  if (tid == 0) s_run = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < n; t0 += TR_NT) {
    // one thread per row for the norm would read 2w strided doubles: a wavefront per row instead, 64 rows a round
    int k = 0;
    for (int r = 0; r < 64; ++r) {
      const int64_t t = t0 + 64 * wv + r;   // wave wv owns rows t0 + 64 wv .. + 63; lane r keeps row r's flag
      double s = 0.0;
      if (t < n) {
        for (int c = lane; c < w; c += 64) s += fabs(xd[t * w + c]) + fabs(yd[t * w + c]);
      }
      s = kwy_wave_sum(s);
      if (lane == r) k = (t < n && !(s < eps)) ? 1 : 0;
    }
    int tot;
    const int off = tr_block_exscan(k, sh, &tot);   // thread (wv, lane) <-> row t0 + 64 wv + lane = t0 + tid
    const int run = s_run;
    s_dst[tid] = k ? run + off : -1;
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
      const int64_t t = t0 + 64 * wv + r;
      const int dst = s_dst[64 * wv + r];
      if (WRITE && t < n && dst >= 0) {
        double *o = joint + (size_t)dst * 2 * w;
        for (int c = lane; c < w; c += 64) { o[c] = xd[t * w + c]; o[w + c] = yd[t * w + c]; }
      }
    }
    __syncthreads();
    if (tid == 0) s_run = run + tot;
    __syncthreads();
  }
  if (tid == 0 && n_out) n_out[0] = s_run;
}

// ---- the kernels: single calls and batches (blockIdx.y or .x = job, descriptors by value) ------------------------
#define TR_BATCH 32          // utterances per launch of the small per-utterance kernels
#define TR_PAIRS 16          // pairs per launch of the row kernels (their views are 128 bytes: 4 KB of arguments at most)
template <class JOB, int N = TR_BATCH>
struct tr_jobs { int n; JOB j[N]; };

struct tr_trim_view { const double *sp; int64_t T; double *rowsum; int64_t *n_out; };
__global__ __launch_bounds__(TR_NT) void k_tr_rowsum(tr_jobs<tr_trim_view> b, int K) {
  const tr_trim_view &q = b.j[blockIdx.y];
  tr_rowsum_body(q.sp, q.T, K, q.rowsum);
}
__global__ __launch_bounds__(TR_NT) void k_tr_trim(tr_jobs<tr_trim_view> b, double eps) {
  const tr_trim_view &q = b.j[blockIdx.x];
  tr_trim_body(q.rowsum, q.T, eps, q.n_out);
}
__global__ __launch_bounds__(TR_NT) void k_tr_align_even(const int32_t *__restrict__ path,
                                                        const int64_t *__restrict__ path_len,
                                                        const double *__restrict__ fx, const double *__restrict__ fy,
                                                        int width, int strict, int use_power, int use_vuv,
                                                        int64_t Tx, int64_t Ty, int pad_len, int32_t *__restrict__ pos,
                                                        int32_t *__restrict__ idx_x, int32_t *__restrict__ idx_y,
                                                        int64_t cap, int64_t *__restrict__ n_out) {
  tr_align_even_body(path, path_len, fx, fy, width, strict, use_power, use_vuv, Tx, Ty, pad_len, pos, idx_x, idx_y, cap, n_out);
}
__global__ __launch_bounds__(TR_NT) void k_tr_joint(const double *__restrict__ xd, const double *__restrict__ yd,
                                                   const int64_t *__restrict__ n_p, int64_t cap, int w, double eps,
                                                   double *__restrict__ joint, int64_t *__restrict__ n_out) {
  tr_joint_body<true>(xd, yd, n_p, cap, w, eps, joint, n_out);
}

// pad_silence's cheap parts for the first n frames of an analysed utterance, stored with pad_len frames of room on
// both ends (kwiiyatta/vocoder/feature.py:19-41; the pad SPECTRA are the generator's business):
//   f0_pad[t] = f0[t - pad] inside [pad, pad + n), 0 elsewhere; aperiodicity rows [pad + n, n + 2 pad) = 1 - 1e-12;
//   voiced[t] = WorldSynthesizer.extract_is_voiced on the padded feature (world.py:147-151)
struct tr_pad_view { const double *f0; int64_t n; double *f0_pad; double *ap_pad; double *voiced; };
__global__ __launch_bounds__(TR_NT) void k_tr_pad(tr_jobs<tr_pad_view> b, int K, int pad, double lowest_f0) {
  const tr_pad_view &q = b.j[blockIdx.y];
  const int64_t Tp = q.n + 2 * pad;
  const double one = 1.0 - 1e-12;
  // tail rows of the aperiodicity: workgroups 0 .. pad - 1 take one row each
  if ((int)blockIdx.x < pad) {
    double *row = q.ap_pad + (size_t)(pad + q.n + blockIdx.x) * K;
    for (int k = threadIdx.x; k < K; k += TR_NT) row[k] = one;
  }
  for (int64_t t = (int64_t)blockIdx.x * TR_NT + threadIdx.x; t < Tp; t += (int64_t)gridDim.x * TR_NT) {
    const bool in = t >= pad && t < pad + q.n;
    const double f = in ? q.f0[t - pad] : 0.0;
    q.f0_pad[t] = f;
    if (q.voiced) {
      const double a0 = in ? q.ap_pad[(size_t)t * K] : one;
      q.voiced[t] = (f >= lowest_f0 && a0 <= 0.999) ? 1.0 : 0.0;
    }
  }
}

// One aligned pair -> its joint rows (include/kwy.h: kwy_train_rows_batch_dev).
struct tr_rows_view {
  kwy_train_job q;
  int32_t *pos, *idx_x, *idx_y;      // scratch: path cells (cap)
  int64_t *n_sel;                    // cells after the filter and the cut
  double *xd, *yd;                   // cap x 3 d
  int64_t cap;
};
__global__ __launch_bounds__(TR_NT) void k_tr_rows_align(tr_jobs<tr_rows_view, TR_PAIRS> b, int width, int strict, int use_power,
                                                        int use_vuv, int pad_len) {
  const tr_rows_view &v = b.j[blockIdx.x];
  tr_align_even_body(v.q.path, v.q.path_len, v.q.feat_x, v.q.feat_y, width, strict, use_power, use_vuv, v.q.x_length,
                     v.q.y_length, pad_len, v.pos, v.idx_x, v.idx_y, v.cap, v.n_sel);
}
// delta features of the selected rows' static coefficients (c0 dropped), both sides: blockIdx.y = 2 pair + side
__global__ void k_tr_rows_delta(tr_jobs<tr_rows_view, TR_PAIRS> b, int d) {
  const tr_rows_view &v = b.j[blockIdx.y >> 1];
  const int side = blockIdx.y & 1;
  const int64_t n = min(v.n_sel[0], v.cap);
  const int64_t t = blockIdx.x;
  if (t >= n) return;
  const double *__restrict__ mc = side ? v.q.mc_y : v.q.mc_x;
  const int32_t *__restrict__ idx = side ? v.idx_y : v.idx_x;
  const int64_t rows = side ? v.q.y_length : v.q.x_length;
  auto row = [&](int64_t tt) { int64_t r = idx[tt]; r = r < 0 ? 0 : (r >= rows ? rows - 1 : r); return mc + r * (d + 1) + 1; };
  const double *r0 = row(t), *rm = t > 0 ? row(t - 1) : r0, *rp = t + 1 < n ? row(t + 1) : r0;
  double *o = (side ? v.yd : v.xd) + t * 3 * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const double xm = t > 0 ? rm[c] : 0.0, x0 = r0[c], xp = t + 1 < n ? rp[c] : 0.0;
    o[c] = x0;
    o[d + c] = (-0.5 * xm + 0.0 * x0) + 0.5 * xp;
    o[2 * d + c] = (1.0 * xm + -2.0 * x0) + 1.0 * xp;
  }
}
__global__ __launch_bounds__(TR_NT) void k_tr_rows_count(tr_jobs<tr_rows_view, TR_PAIRS> b, int w, double eps) {
  const tr_rows_view &v = b.j[blockIdx.x];
  tr_joint_body<false>(v.xd, v.yd, v.n_sel, v.cap, w, eps, nullptr, v.q.n_rows);
}
// the pairs' places in the matrix, in pair order behind the cursor; a pair that would not fit is dropped whole and
// flagged (n_rows = -1 - rows it had)
__global__ void k_tr_rows_place(tr_jobs<tr_rows_view, TR_PAIRS> b, int64_t *__restrict__ cursor, int64_t capacity_rows,
                                int64_t *__restrict__ places) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t at = cursor[0];
  for (int k = 0; k < b.n; ++k) {
    const int64_t n = b.j[k].q.n_rows[0];
    if (at + n > capacity_rows) { places[k] = -1; b.j[k].q.n_rows[0] = -1 - n; continue; }
    places[k] = at;
    at += n;
  }
  cursor[0] = at;
}
__global__ __launch_bounds__(TR_NT) void k_tr_rows_write(tr_jobs<tr_rows_view, TR_PAIRS> b, int w, double eps,
                                                        double *__restrict__ joint, const int64_t *__restrict__ places) {
  const tr_rows_view &v = b.j[blockIdx.x];
  if (places[blockIdx.x] < 0) return;
  tr_joint_body<true>(v.xd, v.yd, v.n_sel, v.cap, w, eps, joint + (size_t)places[blockIdx.x] * 2 * w, nullptr);
}

// ---- C ABI ---------------------------------------------------------------------------------------------
extern "C" int kwy_trim_length_batch_dev(kwy_ctx *ctx, const kwy_trim_job *jobs, int count, int K, double eps) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || K <= 0) { ctx->err = "trim_length: bad argument"; return KWY_EINVAL; }
  size_t bytes = 0;
  for (int j = 0; j < count; ++j) {
    if (!jobs[j].sp || !jobs[j].n_out || jobs[j].T <= 0) { ctx->err = "trim_length: bad argument"; return KWY_EINVAL; }
    bytes += kwy_pad(sizeof(double) * jobs[j].T);
  }
  if (count == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  for (int j0 = 0; j0 < count; j0 += TR_BATCH) {
    tr_jobs<tr_trim_view> b;
    b.n = std::min(TR_BATCH, count - j0);
    int64_t T = 0;
    for (int k = 0; k < b.n; ++k) {
      const kwy_trim_job &q = jobs[j0 + k];
      b.j[k] = tr_trim_view{q.sp, q.T, kwy_arena<double>(ctx, q.T), q.n_out};
      if (!b.j[k].rowsum) { ctx->err = "trim_length: scratch"; return KWY_ENOMEM; }
      T = std::max(T, q.T);
    }
    for (int k = b.n; k < TR_BATCH; ++k) b.j[k] = b.j[0];
    hipLaunchKernelGGL(k_tr_rowsum, dim3((unsigned)((T + 3) / 4), b.n), dim3(TR_NT), 0, ctx->stream, b, K);
    hipLaunchKernelGGL(k_tr_trim, dim3(b.n), dim3(TR_NT), 0, ctx->stream, b, eps);
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_trim_length_dev(kwy_ctx *ctx, const double *sp, int64_t T, int K, double eps, int64_t *n_out) {
  const kwy_trim_job job = {sp, T, n_out};
  return kwy_trim_length_batch_dev(ctx, &job, 1, K, eps);
}

extern "C" int kwy_train_pad_batch_dev(kwy_ctx *ctx, const kwy_pad_job *jobs, int count, int K, int fs, int pad_len) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || K < 2 || fs <= 0 || pad_len < 0) { ctx->err = "train_pad: bad argument"; return KWY_EINVAL; }
  for (int j = 0; j < count; ++j)
    if (!jobs[j].f0 || !jobs[j].f0_pad || !jobs[j].ap_pad || jobs[j].n < 0) { ctx->err = "train_pad: bad argument"; return KWY_EINVAL; }
  if (count == 0) return KWY_OK;
  KWY_HIP(hipSetDevice(ctx->device));
  const double lowest = fs / ((K - 1) / 2.0) + 1.0;
  for (int j0 = 0; j0 < count; j0 += TR_BATCH) {
    tr_jobs<tr_pad_view> b;
    b.n = std::min(TR_BATCH, count - j0);
    int64_t Tp = 0;
    for (int k = 0; k < TR_BATCH; ++k) {
      const kwy_pad_job &q = jobs[j0 + (k < b.n ? k : 0)];
      b.j[k] = tr_pad_view{q.f0, q.n, q.f0_pad, q.ap_pad, q.voiced};
      Tp = std::max(Tp, q.n + 2 * (int64_t)pad_len);
    }
    const unsigned g = (unsigned)std::max<int64_t>(pad_len, (Tp + TR_NT - 1) / TR_NT);
    hipLaunchKernelGGL(k_tr_pad, dim3(g, b.n), dim3(TR_NT), 0, ctx->stream, b, K, pad_len, lowest);
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_train_rows_batch_dev(kwy_ctx *ctx, const kwy_train_job *jobs, int count, int d, int strict,
                                        int use_power, int use_vuv, int pad_len, double eps, double *joint,
                                        int64_t capacity_rows, int64_t *cursor) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0 || d <= 0 || pad_len < 0 || !joint || capacity_rows < 0 || !cursor) {
    ctx->err = "train_rows: bad argument";
    return KWY_EINVAL;
  }
  size_t bytes = 0;
  for (int j = 0; j < count; ++j) {
    const kwy_train_job &q = jobs[j];
    if (!q.path || !q.path_len || !q.feat_x || !q.feat_y || !q.mc_x || !q.mc_y || !q.n_rows || q.x_length <= 0 ||
        q.y_length <= 0) {
      ctx->err = "train_rows: bad argument";
      return KWY_EINVAL;
    }
    const size_t cap = (size_t)(q.x_length + q.y_length + 4);
    bytes += 3 * kwy_pad(sizeof(int32_t) * cap) + kwy_pad(64) + 2 * kwy_pad(sizeof(double) * cap * 3 * d);
  }
  if (count == 0) return KWY_OK;
  bytes += kwy_pad(sizeof(int64_t) * TR_PAIRS) * (size_t)((count + TR_PAIRS - 1) / TR_PAIRS);
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  for (int j0 = 0; j0 < count; j0 += TR_PAIRS) {
    tr_jobs<tr_rows_view, TR_PAIRS> b;
    b.n = std::min(TR_PAIRS, count - j0);
    int64_t cap_max = 0;
    for (int k = 0; k < b.n; ++k) {
      tr_rows_view &v = b.j[k];
      v.q = jobs[j0 + k];
      v.cap = v.q.x_length + v.q.y_length + 4;
      v.pos = kwy_arena<int32_t>(ctx, v.cap);
      v.idx_x = kwy_arena<int32_t>(ctx, v.cap);
      v.idx_y = kwy_arena<int32_t>(ctx, v.cap);
      v.n_sel = kwy_arena<int64_t>(ctx, 8);
      v.xd = kwy_arena<double>(ctx, (size_t)v.cap * 3 * d);
      v.yd = kwy_arena<double>(ctx, (size_t)v.cap * 3 * d);
      if (!v.pos || !v.idx_x || !v.idx_y || !v.n_sel || !v.xd || !v.yd) { ctx->err = "train_rows: scratch"; return KWY_ENOMEM; }
      cap_max = std::max(cap_max, v.cap);
    }
    for (int k = b.n; k < TR_PAIRS; ++k) b.j[k] = b.j[0];
    int64_t *places = kwy_arena<int64_t>(ctx, TR_PAIRS);
    if (!places) { ctx->err = "train_rows: scratch"; return KWY_ENOMEM; }
    hipLaunchKernelGGL(k_tr_rows_align, dim3(b.n), dim3(TR_NT), 0, ctx->stream, b, d + 2, strict, use_power, use_vuv, pad_len);
    hipLaunchKernelGGL(k_tr_rows_delta, dim3((unsigned)cap_max, 2 * b.n), dim3(64), 0, ctx->stream, b, d);
    hipLaunchKernelGGL(k_tr_rows_count, dim3(b.n), dim3(TR_NT), 0, ctx->stream, b, 3 * d, eps);
    hipLaunchKernelGGL(k_tr_rows_place, dim3(1), dim3(64), 0, ctx->stream, b, cursor, capacity_rows, places);
    hipLaunchKernelGGL(k_tr_rows_write, dim3(b.n), dim3(TR_NT), 0, ctx->stream, b, 3 * d, eps, joint, places);
  }
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_is_voiced_dev(kwy_ctx *ctx, const double *f0, const double *ap, int64_t T, int K, int fs,
                                 double *voiced) {
  if (!ctx) return KWY_EINVAL;
  if (!f0 || !ap || !voiced || T <= 0 || K < 2 || fs <= 0) { ctx->err = "is_voiced: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  const double lowest = fs / ((K - 1) / 2.0) + 1.0;
  hipLaunchKernelGGL(k_tr_is_voiced, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, ctx->stream, f0, ap, T, K, lowest,
                     voiced);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_align_even_dev(kwy_ctx *ctx, const int32_t *path, const int64_t *path_len, const double *feat_x,
                                  const double *feat_y, int width, int strict, int use_power, int use_vuv, int64_t Tx,
                                  int64_t Ty, int pad_len, int32_t *idx_x, int32_t *idx_y, int64_t capacity,
                                  int64_t *n_out) {
  if (!ctx) return KWY_EINVAL;
  if (!path || !path_len || !feat_x || !feat_y || !idx_x || !idx_y || !n_out || width < 2 || Tx <= 0 || Ty <= 0 ||
      pad_len < 0 || capacity <= 0) {
    ctx->err = "align_even: bad argument";
    return KWY_EINVAL;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  KWY_TRY(kwy_arena_begin(ctx, kwy_pad(sizeof(int32_t) * (size_t)(Tx + Ty + 4))));
  int32_t *pos = kwy_arena<int32_t>(ctx, (size_t)(Tx + Ty + 4));
  if (!pos) { ctx->err = "align_even: scratch"; return KWY_ENOMEM; }
  hipLaunchKernelGGL(k_tr_align_even, dim3(1), dim3(TR_NT), 0, ctx->stream, path, path_len, feat_x, feat_y, width,
                     strict, use_power, use_vuv, Tx, Ty, pad_len, pos, idx_x, idx_y, capacity, n_out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_delta_features_dev(kwy_ctx *ctx, const double *x, const int64_t *n, int64_t capacity, int d,
                                      double *out) {
  if (!ctx) return KWY_EINVAL;
  if (!x || !n || !out || capacity <= 0 || d <= 0) { ctx->err = "delta_features: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_tr_delta, dim3((unsigned)capacity), dim3(64), 0, ctx->stream, x, n, capacity, d, out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_joint_rows_dev(kwy_ctx *ctx, const double *xd, const double *yd, const int64_t *n, int64_t capacity,
                                  int width, double eps, double *joint, int64_t *n_out) {
  if (!ctx) return KWY_EINVAL;
  if (!xd || !yd || !n || !joint || !n_out || capacity <= 0 || width <= 0) { ctx->err = "joint_rows: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_tr_joint, dim3(1), dim3(TR_NT), 0, ctx->stream, xd, yd, n, capacity, width, eps, joint, n_out);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}
