// kwy_finish.hip -- the post-step of a synthesised waveform and its 16-bit PCM, on the device.
//
// Replaces, for a batch of waveforms that are already in HBM, what the reference does on the host per file after
// pyworld.synthesize:
//   Synthesizer.synthesize(normalize=True)   kwiiyatta/vocoder/abc/synthesizer.py:11-20
//       wavdata.normalize(None)              data -= data.mean()
//       for i in range(feature.frame_len):   normalize_data(data[fs*i//1000 : fs*(i+1)//1000], peak_lv)
//   Wavdata.save(normalize=True)             kwiiyatta/wavfile.py:8-29
//       data -= data.mean(); normalize_data(data, peak_lv); (data * 2**15).astype(np.int16)
// so that a wave of utterances downloads 2 bytes per sample instead of 8 and no numpy slice loop (~2 000 calls per
// 10 s utterance) runs between the GPU and the wav file.
//
// The int16 samples equal the host path's exactly, which needs numpy's mean to the last bit: np.add.reduce over a
// contiguous float64 array adds, in order, the sums of consecutive chunks of 8192 elements (np.getbufsize()), each
// chunk summed by numpy's pairwise routine (DOUBLE_pairwise_sum: blocks of <= 128 elements with eight interleaved
// accumulators, halves split at a multiple of 8).  k_fin_chunk_sums reproduces that tree: the 64 threads of a
// wavefront descend to "their" node at depth 6 (a full chunk: 64 leaves of 128 elements), evaluate it with the same
// recursion, and the tree above is combined level by level.  Everything after the means is exact arithmetic
// (max |x|, one division, products, truncation).
#include <math.h>

#include "kwy_internal.hpp"

#define FIN_CHUNK 8192     // np.getbufsize()
#define FIN_LEAF 128       // numpy's PW_BLOCKSIZE
#define FIN_MS 64          // 1 ms pieces per workgroup of the limiting kernel

struct fin_utt {
  const double *y;   // n samples (synthesis output)
  double *work;      // n samples of scratch: the waveform after the post-step
  double *sums;      // 2 * nchunks_max chunk sums (first and second mean), then 1 + nchunks_max peaks
  int16_t *pcm;      // n samples out
  int64_t n;
  int64_t frame_len; // pieces of 1 ms that get limited (the feature's frame count)
};

struct fin_plan {
  int count, fs, nchunks_max;
  int do_synth, do_save;       // the two normalisation steps (normalize=True of synthesize / of save)
  int limit_save;              // save's peak limit (peak_lv is not None)
  double ceil_piece, ceil_save;
  fin_utt u[KWY_BATCH_MAX];
};

// numpy's leaf: n <= 128
__device__ inline double fin_leaf(const double *__restrict__ a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += a[i];
  return res;
}

// numpy's recursion below depth 6 of a chunk: such a node has at most 128 + 15 elements (a right child gets at most
// half of its parent plus 7.5), i.e. it is a leaf or splits once more into two leaves
__device__ inline double fin_node(const double *__restrict__ a, int n) {
  if (n <= FIN_LEAF) return fin_leaf(a, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return fin_leaf(a, n2) + fin_leaf(a + n2, n - n2);
}

// sums[slot * nchunks_max + c] = numpy's pairwise sum of chunk c of `src` (one wavefront per chunk)
template <bool FROM_WORK>
__global__ __launch_bounds__(64) void k_fin_chunk_sums(fin_plan P, int slot) {
  __shared__ double v[64];
  __shared__ int ok[64];
  const int utt = blockIdx.y, lane = threadIdx.x;
  const fin_utt &U = P.u[utt];
  const int64_t c0 = (int64_t)blockIdx.x * FIN_CHUNK;
  if (c0 >= U.n) return;
  const double *__restrict__ a = (FROM_WORK ? U.work : U.y) + c0;
  const int L = (int)(U.n - c0 < FIN_CHUNK ? U.n - c0 : FIN_CHUNK);
  // descend six levels by the bits of the lane (most significant first); an early leaf belongs to the lane whose
  // remaining bits are zero
  int start = 0, len = L, valid = 1;
  for (int d = 5; d >= 0; --d) {
    if (len <= FIN_LEAF) {
      if (lane & ((2 << d) - 1)) valid = 0;
      break;
    }
    int n2 = len / 2;
    n2 -= n2 % 8;
    if ((lane >> d) & 1) { start += n2; len -= n2; } else { len = n2; }
  }
  v[lane] = valid ? fin_node(a + start, len) : 0.0;
  ok[lane] = valid;
  __syncthreads();
  // the tree above: node (level, i) = left + right when the right child exists, else the left child as it is
  for (int w = 1; w < 64; w <<= 1) {
    double s = 0.0;
    int keep = 0;
    if ((lane & (2 * w - 1)) == 0) {
      s = v[lane];
      keep = ok[lane];
      if (ok[lane + w]) s = s + v[lane + w];
    }
    __syncthreads();
    if ((lane & (2 * w - 1)) == 0) { v[lane] = s; ok[lane] = keep; }
    __syncthreads();
  }
  if (lane == 0) U.sums[(int64_t)slot * P.nchunks_max + blockIdx.x] = v[0];
}

// the mean numpy computes: chunk sums added in order (starting from 0.0) / n
__device__ __forceinline__ double fin_mean(const fin_plan &P, const fin_utt &U, int slot) {
  const int nch = (int)((U.n + FIN_CHUNK - 1) / FIN_CHUNK);
  const double *__restrict__ s = U.sums + (int64_t)slot * P.nchunks_max;
  double acc = 0.0;
  for (int i = 0; i < nch; ++i) acc += s[i];
  return acc / (double)U.n;
}

// work = y - mean, then the peak limit of the 1 ms pieces i < frame_len.  Workgroup b owns the pieces
// [64 b, 64 b + 64): samples [fs*64b//1000, fs*64(b+1)//1000); the last one also the rest of the waveform.
__global__ __launch_bounds__(KWY_THREADS) void k_fin_pieces(fin_plan P) {
  extern __shared__ double tile[];
  const int utt = blockIdx.y, tid = threadIdx.x;
  const fin_utt &U = P.u[utt];
  const int64_t fs = P.fs;
  const int64_t p0 = (int64_t)blockIdx.x * FIN_MS;
  int64_t s_lo = fs * p0 / 1000, s_hi = fs * (p0 + FIN_MS) / 1000;
  if (s_lo >= U.n) return;
  if (s_hi > U.n) s_hi = U.n;          // (the groups of a launch cover every sample of its longest waveform)
  const double mean = P.do_synth ? fin_mean(P, U, 0) : 0.0;
  const int m = (int)(s_hi - s_lo);
  for (int i = tid; i < m; i += KWY_THREADS) tile[i] = U.y[s_lo + i] - mean;
  __syncthreads();
  if (P.do_synth) {
    const int wv = tid >> 6, lane = tid & 63;
    for (int q = wv; q < FIN_MS; q += KWY_WAVES) {
      const int64_t piece = p0 + q;
      if (piece >= U.frame_len) break;
      const int a = (int)(fs * piece / 1000 - s_lo), b = (int)(fs * (piece + 1) / 1000 - s_lo);
      double pk = 0.0;
      for (int i = a + lane; i < b; i += 64) pk = fmax(pk, fabs(tile[i]));
      pk = kwy_wave_max_f64(pk);
      if (pk > P.ceil_piece) {
        const double g = P.ceil_piece / pk;
        for (int i = a + lane; i < b; i += 64) tile[i] *= g;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < m; i += KWY_THREADS) U.work[s_lo + i] = tile[i];
}

// per chunk: max |work - mean2|
__global__ __launch_bounds__(KWY_THREADS) void k_fin_peaks(fin_plan P) {
  __shared__ double red[8];
  const int utt = blockIdx.y, tid = threadIdx.x;
  const fin_utt &U = P.u[utt];
  const int64_t c0 = (int64_t)blockIdx.x * FIN_CHUNK;
  if (c0 >= U.n) return;
  const double mean = fin_mean(P, U, 1);
  const int L = (int)(U.n - c0 < FIN_CHUNK ? U.n - c0 : FIN_CHUNK);
  double pk = 0.0;
  for (int i = tid; i < L; i += KWY_THREADS) pk = fmax(pk, fabs(U.work[c0 + i] - mean));
  pk = kwy_wave_max_f64(pk);
  if ((tid & 63) == 0) red[tid >> 6] = pk;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < KWY_WAVES; ++w) pk = fmax(pk, red[w]);
    U.sums[2 * (int64_t)P.nchunks_max + blockIdx.x] = pk;
  }
}

// pcm = (int16) trunc(((work - mean2) * gain) * 2^15)
__global__ __launch_bounds__(KWY_THREADS) void k_fin_pcm(fin_plan P) {
  const int utt = blockIdx.y, tid = threadIdx.x;
  const fin_utt &U = P.u[utt];
  const int64_t c0 = (int64_t)blockIdx.x * FIN_CHUNK;
  if (c0 >= U.n) return;
  double mean = 0.0, gain = 1.0;
  bool scaled = false;
  if (P.do_save) {
    mean = fin_mean(P, U, 1);
    if (P.limit_save) {
      const int nch = (int)((U.n + FIN_CHUNK - 1) / FIN_CHUNK);
      double pk = 0.0;
      for (int i = 0; i < nch; ++i) pk = fmax(pk, U.sums[2 * (int64_t)P.nchunks_max + i]);
      if (pk > P.ceil_save) { gain = P.ceil_save / pk; scaled = true; }
    }
  }
  const int L = (int)(U.n - c0 < FIN_CHUNK ? U.n - c0 : FIN_CHUNK);
  for (int i = tid; i < L; i += KWY_THREADS) {
    double v = U.work[c0 + i] - mean;
    if (scaled) v *= gain;
    v *= 32768.0;
    // numpy's astype(int16) of a float64: the C conversion through a wider integer, low 16 bits kept
    // (cvttsd2si semantics: beyond the 32-bit range, and for NaN, the "integer indefinite" value, whose low half is 0)
    const int32_t w = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000;
    U.pcm[c0 + i] = (int16_t)w;
  }
}

extern "C" int64_t kwy_finish_scratch_bytes(int64_t y_length) {
  if (y_length <= 0) return 0;
  const int64_t nch = (y_length + FIN_CHUNK - 1) / FIN_CHUNK;
  return (int64_t)(kwy_pad(sizeof(double) * (size_t)y_length) + kwy_pad(sizeof(double) * (size_t)(3 * nch + 1)));
}

extern "C" int kwy_finish_pcm16_batch_dev(kwy_ctx *ctx, const kwy_finish_job *jobs, int count, int fs,
                                          int normalize_synth, double piece_ceiling, int normalize_save,
                                          double save_ceiling) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 1 || fs <= 0 || (normalize_synth && !(piece_ceiling > 0))) {
    ctx->err = "finish_pcm16: bad argument";
    return KWY_EINVAL;
  }
  int64_t n_max = 0;
  for (int i = 0; i < count; ++i) {
    const kwy_finish_job &q = jobs[i];
    if (!q.y || !q.pcm || q.y_length <= 0 || q.y_length > 0x3fffffff || q.frame_len < 0) {
      ctx->err = "finish_pcm16: bad argument";
      return KWY_EINVAL;
    }
    // np.abs(piece).max() of an empty slice raises in the reference's loop
    if (normalize_synth && q.frame_len > 0 && (int64_t)fs * (q.frame_len - 1) / 1000 >= q.y_length) {
      ctx->err = "finish_pcm16: more frames than milliseconds of waveform (the reference's loop fails on an empty piece)";
      return KWY_EINVAL;
    }
    n_max = q.y_length > n_max ? q.y_length : n_max;
  }
  KWY_HIP(hipSetDevice(ctx->device));
  const int per_pass = count < KWY_BATCH_MAX ? count : KWY_BATCH_MAX;
  const size_t per_utt = (size_t)kwy_finish_scratch_bytes(n_max);
  KWY_TRY(kwy_arena_begin(ctx, per_utt * per_pass));
  char *scratch = (char *)kwy_arena_alloc(ctx, per_utt * per_pass);
  if (!scratch) { ctx->err = "finish_pcm16: scratch arena too small"; return KWY_ENOMEM; }
  fin_plan P;
  P.fs = fs;
  P.nchunks_max = (int)((n_max + FIN_CHUNK - 1) / FIN_CHUNK);
  P.do_synth = normalize_synth ? 1 : 0;
  P.do_save = normalize_save ? 1 : 0;
  P.limit_save = (normalize_save && save_ceiling == save_ceiling && save_ceiling > 0) ? 1 : 0;   // NaN: peak_lv=None
  P.ceil_piece = piece_ceiling;
  P.ceil_save = save_ceiling;
  const size_t tile_bytes = sizeof(double) * (size_t)((int64_t)fs * FIN_MS / 1000 + 2);
  if (tile_bytes > 150 * 1024) { ctx->err = "finish_pcm16: sampling rate too high"; return KWY_EINVAL; }
  KWY_HIP(hipFuncSetAttribute((const void *)k_fin_pieces, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
  for (int i0 = 0; i0 < count; i0 += KWY_BATCH_MAX) {
    const int m = count - i0 < KWY_BATCH_MAX ? count - i0 : KWY_BATCH_MAX;
    P.count = m;
    int64_t ms_max = 0;
    for (int u = 0; u < KWY_BATCH_MAX; ++u) {
      if (u < m) {
        const kwy_finish_job &q = jobs[i0 + u];
        char *base = scratch + per_utt * u;
        P.u[u] = fin_utt{q.y, (double *)base, (double *)(base + kwy_pad(sizeof(double) * (size_t)n_max)), q.pcm,
                         q.y_length, q.frame_len};
        const int64_t ms = (q.y_length * 1000 + fs - 1) / fs + 1;
        ms_max = ms > ms_max ? ms : ms_max;
      } else {
        P.u[u] = fin_utt{nullptr, nullptr, nullptr, nullptr, 0, 0};
      }
    }
    const int groups = (int)((ms_max + FIN_MS - 1) / FIN_MS);
    kwy_prof_scope ps_(ctx, "k_finish");
    if (P.do_synth)
      hipLaunchKernelGGL(k_fin_chunk_sums<false>, dim3(P.nchunks_max, m), dim3(64), 0, ctx->stream, P, 0);
    hipLaunchKernelGGL(k_fin_pieces, dim3(groups, m), dim3(KWY_THREADS), tile_bytes, ctx->stream, P);
    if (P.do_save) {
      hipLaunchKernelGGL(k_fin_chunk_sums<true>, dim3(P.nchunks_max, m), dim3(64), 0, ctx->stream, P, 1);
      if (P.limit_save) hipLaunchKernelGGL(k_fin_peaks, dim3(P.nchunks_max, m), dim3(KWY_THREADS), 0, ctx->stream, P);
    }
    hipLaunchKernelGGL(k_fin_pcm, dim3(P.nchunks_max, m), dim3(KWY_THREADS), 0, ctx->stream, P);
    KWY_HIP(hipGetLastError());
  }
  return KWY_OK;
}
