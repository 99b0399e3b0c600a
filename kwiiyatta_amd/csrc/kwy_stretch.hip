// kwy_stretch.hip -- the spectral-axis stretch of `reshape` on gfx950.
//
// Replaces, for a (T, K) matrix of power spectra or aperiodicity ratios,
//     np.exp(scipy.signal.resample_poly(np.hstack((first bin x pad, np.log(rows), last bin x pad)),
//                                       new_K, K, axis=1)[:, trim:-trim])
// -- Synthesizer._reshape_feature and its two callers, kwiiyatta/vocoder/abc/synthesizer.py:31-54 (pad = 20 periods
// of K / gcd, trim = 20 periods of new_K / gcd), reached from reshape / resample whenever source, target and
// converter rates differ (vocoder/abc/feature.py, vocoder/mcep.py:31-45).
//
// scipy's resample_poly (1.x, window ('kaiser', 5.0), zero extension) is an FIR of 20 max(up, down) + 1 taps run as
// a polyphase up/down filter: y[n] = sum_m x[m] h[n down - m up - n_pre_pad].  The taps depend on (up, down) only
// and are designed here on the host with the same formulas (firwin: windowed sinc scaled to unit DC gain; Kaiser
// window through the power series of I0), cached per context.  The padded row is never materialised: the edge
// replication is an index clamp when a tile of rows is staged (as logarithms) in LDS.
//
// One workgroup = 256 output bins x R rows: a thread owns one output bin, walks "its" polyphase branch of the
// filter once and applies every tap to R rows from LDS, so the tap table (up to 330 KB, L2-resident) is read once
// per R rows.  Algorithmic HBM bytes per row: (K + new_K) * 8.
#include <math.h>

#include <vector>

#include "kwy_internal.hpp"

#define ST_COLS KWY_THREADS
#define ST_MAX_ROWS 8
#define ST_LDS_DOUBLES 8192   // 64 KB of staged logarithms per workgroup

struct stretch_plan {
  int up, down, numtaps, lead_in, n_in;
  long long shift0;   // filter index of input 0 for output bin 0:  k = shift0 + j down - m up
  int span, rows;     // staged inputs per tile and row, rows per workgroup
};

__global__ __launch_bounds__(KWY_THREADS) void k_stretch_log(const double *__restrict__ rows, int64_t T, int K,
                                                            int new_K, const double *__restrict__ h, stretch_plan p,
                                                            double *__restrict__ out) {
  extern __shared__ double xs[];   // [p.rows][p.span]
  const int tid = threadIdx.x;
  const int j0 = blockIdx.x * ST_COLS;
  const int64_t row0 = (int64_t)blockIdx.y * p.rows;
  // inputs the tile's outputs touch: m in [ceil((s_first - numtaps + 1) / up), floor(s_last / up)]
  const long long s_first = p.shift0 + (long long)j0 * p.down;
  const long long a = s_first - p.numtaps + 1;
  const long long m_lo = a >= 0 ? (a + p.up - 1) / p.up : -((-a) / p.up);
  for (int r = 0; r < p.rows; ++r) {
    const int64_t row = row0 + r;
    const double *src = rows + row * K;
    for (int s = tid; s < p.span; s += KWY_THREADS) {
      const long long m = m_lo + s;
      double v = 0.0;
      if (row < T && m >= 0 && m < p.n_in) {
        long long bin = m - p.lead_in;                        // the replicated edges: a clamp
        bin = bin < 0 ? 0 : (bin > K - 1 ? K - 1 : bin);
        v = log(src[bin]);
      }
      xs[r * p.span + s] = v;
    }
  }
  __syncthreads();
  const int j = j0 + tid;
  if (j >= new_K) return;
  const long long s = p.shift0 + (long long)j * p.down;
  long long mt = s / p.up;
  int k = (int)(s - mt * p.up);                               // this output's polyphase branch
  double acc[ST_MAX_ROWS];
#pragma unroll
  for (int r = 0; r < ST_MAX_ROWS; ++r) acc[r] = 0.0;
  for (; k < p.numtaps && mt >= m_lo; k += p.up, --mt) {
    const double hv = h[k];
    const double *col = xs + (int)(mt - m_lo);
#pragma unroll
    for (int r = 0; r < ST_MAX_ROWS; ++r)
      if (r < p.rows) acc[r] += hv * col[r * p.span];
  }
#pragma unroll
  for (int r = 0; r < ST_MAX_ROWS; ++r)
    if (r < p.rows && row0 + r < T) out[(row0 + r) * new_K + j] = exp(acc[r]);
}

// ------------------------------------------------------------------ host side
namespace {

double bessel_i0(double x) {          // power series; x <= 5 here (Kaiser beta): converges in ~25 terms to 1e-17
  const double q = x * x / 4.0;
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * k);
    sum += term;
    if (term < sum * 1e-18) break;
  }
  return sum;
}

// scipy.signal.firwin(numtaps, cutoff, window=('kaiser', 5.0)) with Nyquist = 1, times `gain`
std::vector<double> design_taps(int numtaps, double cutoff, double gain) {
  std::vector<double> h(numtaps);
  const double alpha = 0.5 * (numtaps - 1);
  const double i0b = bessel_i0(5.0);
  double dc = 0.0;
  for (int i = 0; i < numtaps; ++i) {
    const double m = i - alpha;
    const double y = M_PI * (cutoff * m == 0.0 ? 1.0e-20 : cutoff * m);     // numpy.sinc
    const double lowpass = cutoff * (sin(y) / y);
    const double rel = (i - alpha) / alpha;
    const double w = bessel_i0(5.0 * sqrt(fmax(0.0, 1.0 - rel * rel))) / i0b;
    h[i] = lowpass * w;
    dc += h[i];
  }
  for (double &v : h) v = v / dc * gain;
  return h;
}

int gcd_int(int a, int b) {
  while (b) { const int t = a % b; a = b; b = t; }
  return a;
}

}  // namespace

static int stretch_make_plan(kwy_ctx *ctx, int K, int new_K, stretch_plan *p, const double **taps) {
  const int g = gcd_int(K, new_K);
  p->up = new_K / g;
  p->down = K / g;
  p->lead_in = p->down * 20;                 // spectrum_len // gcd * 20 bins replicated on either side
  const int lead_out = p->up * 20;           // ... which are new_spectrum_len // gcd * 20 output bins
  p->n_in = K + 2 * p->lead_in;
  const int max_rate = p->up > p->down ? p->up : p->down;
  const long long half_len = 10LL * max_rate;
  if (2 * half_len + 1 > (1LL << 24)) { ctx->err = "stretch: bin counts too incommensurable"; return KWY_EINVAL; }
  p->numtaps = (int)(2 * half_len + 1);
  const long long n_pre_pad = p->down - half_len % p->down;
  const long long n_pre_remove = (half_len + n_pre_pad) / p->down;
  p->shift0 = (n_pre_remove + lead_out) * p->down - n_pre_pad;
  // staging: the widest tile needs the inputs of 256 consecutive outputs plus one filter length
  const long long span = ((long long)(ST_COLS - 1) * p->down + p->numtaps) / p->up + 3;
  if (span > ST_LDS_DOUBLES) { ctx->err = "stretch: ratio of bin counts too large"; return KWY_EINVAL; }
  p->span = (int)span;
  p->rows = (int)(ST_LDS_DOUBLES / span);
  if (p->rows > ST_MAX_ROWS) p->rows = ST_MAX_ROWS;
  const std::string key = "stretch:" + std::to_string(p->up) + "/" + std::to_string(p->down);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    const std::vector<double> h = design_taps(p->numtaps, 1.0 / max_rate, (double)p->up);
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * h.size()));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  *taps = it->second;
  return KWY_OK;
}

static int stretch_check(kwy_ctx *ctx, const void *rows, int64_t T, int K, int new_K, const void *out) {
  if (!ctx) return KWY_EINVAL;
  if (!rows || !out || T <= 0 || K <= 0 || new_K <= 0) { ctx->err = "stretch: bad argument"; return KWY_EINVAL; }
  return KWY_OK;
}

extern "C" int kwy_stretch_log_dev(kwy_ctx *ctx, const double *rows, int64_t T, int K, int new_K, double *out) {
  KWY_TRY(stretch_check(ctx, rows, T, K, new_K, out));
  KWY_HIP(hipSetDevice(ctx->device));
  if (K == new_K) {                         // resample_poly returns a copy; exp(log(x)) is not x, so neither do we
    KWY_HIP(hipMemcpyAsync(out, rows, sizeof(double) * (size_t)T * K, hipMemcpyDeviceToDevice, ctx->stream));
    return KWY_OK;
  }
  stretch_plan p;
  const double *taps;
  KWY_TRY(stretch_make_plan(ctx, K, new_K, &p, &taps));
  const size_t lds = sizeof(double) * (size_t)p.rows * p.span;
  KWY_HIP(hipFuncSetAttribute((const void *)k_stretch_log, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((unsigned)((new_K + ST_COLS - 1) / ST_COLS), (unsigned)((T + p.rows - 1) / p.rows));
  if (grid.y > 65535u) { ctx->err = "stretch: too many frames for one call"; return KWY_EINVAL; }
  KWY_PROF(ctx, "k_stretch_log", hipLaunchKernelGGL(k_stretch_log, grid, dim3(KWY_THREADS), lds, ctx->stream, rows, T, K,
                                                    new_K, taps, p, out));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

extern "C" int kwy_stretch_log(kwy_ctx *ctx, const double *rows, int64_t T, int K, int new_K, double *out) {
  KWY_TRY(stretch_check(ctx, rows, T, K, new_K, out));
  KWY_HIP(hipSetDevice(ctx->device));
  const size_t bi = kwy_pad(sizeof(double) * (size_t)T * K), bo = kwy_pad(sizeof(double) * (size_t)T * new_K);
  KWY_TRY(kwy_arena_begin(ctx, bi + bo));
  double *din = kwy_arena<double>(ctx, (size_t)T * K), *dout = kwy_arena<double>(ctx, (size_t)T * new_K);
  KWY_HIP(hipMemcpyAsync(din, rows, sizeof(double) * (size_t)T * K, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(kwy_stretch_log_dev(ctx, din, T, K, new_K, dout));
  KWY_HIP(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)T * new_K, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
