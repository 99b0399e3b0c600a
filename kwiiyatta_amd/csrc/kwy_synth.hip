// kwy_synth.hip -- WORLD pulse-synchronous overlap-add synthesis on gfx950.
//
// Replaces pyworld.synthesize (reference call site kwiiyatta/vocoder/world.py:86-92;
// algorithm: Morise et al. 2016 "WORLD", synthesis.cpp as shipped with pyworld 0.2.8).
//
//   time base   : per-sample f0/vuv interpolation, phase accumulation as a
//                 device-wide f64 scan (tile sums -> scan -> tile rescan),
//                 pulse detection + ordered compaction          (5 tiny kernels)
//   noise       : pulse p starts (idx[p]-idx[0]) draws into the serial randn
//                 stream -> read from the device's table of that stream
//                 (jump-ahead inside the pulse kernel beyond the table)
//   per pulse   : one 256-thread workgroup: interpolate the two neighbouring
//                 spectral/aperiodicity frames (coalesced HBM reads), build the
//                 minimum-phase periodic and aperiodic responses with LDS FFTs,
//                 and overlap-add the 2*(K-1) samples into y with f64 atomics.
//
// Algorithmic HBM bytes per frame: 2*K*8 + 8 in, hop*8 out.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "kwy_internal.hpp"

#define SYN_SAFE 0.000000000001
#define SYN_DEFAULT_F0 500.0
#define SYN_TILE_PER_THREAD 4
#define SYN_TILE (KWY_THREADS * SYN_TILE_PER_THREAD)
#define SYN_TWO_PI (2.0 * KWY_PI)
// Overlap-add without atomics: a pulse's response (fft_size samples) goes to its own slot of a response buffer;
// k_syn_ola then adds, per output sample, the responses that reach it IN PULSE ORDER -- the order of the serial
// reference loop.  Every sum has a fixed order: two runs give the same bits (the reference asserts exactly that,
// tests/kwiiyatta/test_vocoder.py:171).  The buffer holds SYN_SLOTS(y_length) pulses (every signal whose f0 stays
// averages at most 640 Hz fits -- unvoiced stretches run at 500 Hz, DIO's ceiling is 800 Hz); pulses beyond that
// are added afterwards by ONE workgroup, pulse by pulse in the same order (slow, but such signals do not occur in
// speech), so the summation order is the serial one in every case.
#define SYN_SLOTS(y_length, fs) ((int)((y_length) * 640 / (fs)) + 64)

struct syn_params {
  int64_t T, y_length;
  int fs, fft_size;
  double frame_period;  // seconds
  double lowest_f0;
  double sp_mul;
};

// per-sample phase increment and vuv (synthesis.cpp GetTimeBase + interp1/histc)
__device__ __forceinline__ double syn_increment(const double *__restrict__ f0, const syn_params &p,
                                                int64_t n, double *vuv_out) {
  const double fp = p.frame_period;
  const double xi = n / (double)p.fs;
  int64_t g = (int64_t)(xi / fp);
  if (g > p.T) g = p.T;
  while (g + 1 <= p.T && (g + 1) * fp <= xi) ++g;
  while (g > 0 && g * fp > xi) --g;
  int64_t k = g + 1;  // number of nodes <= xi, clamped to [1, T]
  if (k < 1) k = 1;
  if (k > p.T) k = p.T;
  const double xa = (k - 1) * fp, xb = k * fp;
  const double s = (xi - xa) / (xb - xa);
  auto cf = [&](int64_t j) -> double {
    if (j < p.T) { double v = f0[j]; return v < p.lowest_f0 ? 0.0 : v; }
    double a = f0[p.T - 1], b = f0[p.T - 2];
    a = a < p.lowest_f0 ? 0.0 : a;
    b = b < p.lowest_f0 ? 0.0 : b;
    return a * 2 - b;
  };
  auto cv = [&](int64_t j) -> double {
    if (j < p.T) { double v = f0[j]; return v < p.lowest_f0 ? 0.0 : 1.0; }
    double a = f0[p.T - 1] < p.lowest_f0 ? 0.0 : 1.0;
    double b = f0[p.T - 2] < p.lowest_f0 ? 0.0 : 1.0;
    return a * 2 - b;
  };
  const double fa = cf(k - 1), fb = cf(k);
  const double va = cv(k - 1), vb = cv(k);
  double f = fa + s * (fb - fa);
  double v = va + s * (vb - va);
  v = v > 0.5 ? 1.0 : 0.0;
  if (v == 0.0) f = SYN_DEFAULT_F0;
  *vuv_out = v;
  return 2.0 * KWY_PI * f / p.fs;
}

__device__ __forceinline__ int syn_block_exscan_int(int v, int *sh, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  __syncthreads();
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < KWY_WAVES; ++i) {
    if (i < wv) woff += sh[i];
    tot += sh[i];
  }
  *total = tot;
  return woff + (inc - v);
}

__device__ __forceinline__ void syn_scan_counts_body(int *__restrict__ cnt, int nt,
                                                                int *__restrict__ npulse) {
  __shared__ int tot[KWY_THREADS];
  const int t = threadIdx.x;
  const int chunk = (nt + KWY_THREADS - 1) / KWY_THREADS;
  const int b0 = t * chunk, b1 = min(nt, b0 + chunk);
  int run = 0;
  for (int i = b0; i < b1; ++i) run += cnt[i];
  tot[t] = run;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < KWY_THREADS; ++i) { int v = tot[i]; tot[i] = acc; acc += v; }
    npulse[0] = acc;
  }
  __syncthreads();
  run = tot[t];
  for (int i = b0; i < b1; ++i) { int v = cnt[i]; cnt[i] = run; run += v; }
}

// phase A: per-sample phase increment 2*pi*f0/fs and vuv flag
__device__ __forceinline__ void syn_inc_body(const double *__restrict__ f0, syn_params p,
                                                        double *__restrict__ inc,
                                                        unsigned char *__restrict__ vuv8) {
  const int64_t n = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x;
  if (n >= p.y_length) return;
  double vuv;
  inc[n] = syn_increment(f0, p, n, &vuv);
  vuv8[n] = vuv > 0.5 ? 1 : 0;
}

// phase B: the running phase  tp[n] = fl(tp[n-1] + inc[n])  EXACTLY as the serial
// CPU loop rounds it, but computed with block-wide scans.
//
// While tp stays inside one binade [2^k, 2^(k+1)) every addition rounds to a
// multiple of u = 2^(k-52):  fl(m*u + c) = (m + rn(c/u))*u, where rn is
// round-to-nearest-even on the integer m + c/u.  Only an exact tie
// (frac(c/u) == 1/2) depends on the running value, and then only on the parity
// of m.  So each sample is a map {parity in} -> {integer delta}; such maps
// compose associatively and can be scanned.  The few additions that carry tp
// into the next binade are done with a real f64 add, after which the scan
// restarts in the new binade.  (Pulse positions are decided by these roundings
// whenever the period is an exact number of samples, e.g. the 500 Hz default of
// unvoiced segments at 16/48 kHz, so this is needed for parity.)
#define SYN_PH_THREADS 1024
#define SYN_PH_PER_THREAD 4
#define SYN_PH_TILE (SYN_PH_THREADS * SYN_PH_PER_THREAD)

struct syn_ff { long long d0, d1; };  // delta when the incoming mantissa is even / odd

__device__ __forceinline__ syn_ff syn_ff_compose(syn_ff g, syn_ff f) {  // g first, then f
  syn_ff h;
  h.d0 = g.d0 + ((g.d0 & 1) ? f.d1 : f.d0);
  h.d1 = g.d1 + (((g.d1 + 1) & 1) ? f.d1 : f.d0);
  return h;
}

// c / 2^(k-52) split into integer part + rounding class, as a parity map.
__device__ __forceinline__ syn_ff syn_ff_make(double c, int k) {
  syn_ff r = {0, 0};
  if (!(c > 0.0)) return r;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(c);
  const int e = (int)((bits >> 52) & 0x7ff) - 1023;
  const unsigned long long mant = (bits & 0xfffffffffffffULL) | 0x10000000000000ULL;
  const int s = k - e;  // c/u = mant * 2^(-s)
  if (s <= 0) {
    long long v = (s > -10) ? (long long)(mant << (-s)) : (1LL << 62);  // forces the binade check to trip
    r.d0 = r.d1 = v;
    return r;
  }
  if (s > 54) return r;  // c < u/4: never moves the sum
  if (s == 54) return r; // mant < 2^53 = half of 2^54: below half
  const long long kint = (long long)(mant >> s);
  const unsigned long long rem = mant & ((1ULL << s) - 1ULL);
  const unsigned long long half = 1ULL << (s - 1);
  if (rem < half) { r.d0 = r.d1 = kint; }
  else if (rem > half) { r.d0 = r.d1 = kint + 1; }
  else { r.d0 = kint + (kint & 1); r.d1 = kint + ((kint + 1) & 1); }  // tie: to even
  return r;
}

// ordered exclusive scan of one map per thread over the workgroup (NW waves);
// returns the composition of all earlier threads' maps, *total = the whole block's.
// wtot: 2 NW + 1 maps of LDS.  (The waves' totals are scanned once, by the first wave, not by every thread.)
template <int NW>
__device__ __forceinline__ syn_ff syn_ff_block_exscan(syn_ff mine, syn_ff *wtot, syn_ff *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  syn_ff incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    syn_ff u;
    u.d0 = __shfl_up(incl.d0, o);
    u.d1 = __shfl_up(incl.d1, o);
    if (lane >= o) incl = syn_ff_compose(u, incl);
  }
  __syncthreads();
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  if (wv == 0) {
    syn_ff w = lane < NW ? wtot[lane] : syn_ff{0, 0};
#pragma unroll
    for (int o = 1; o < NW; o <<= 1) {
      syn_ff u;
      u.d0 = __shfl_up(w.d0, o);
      u.d1 = __shfl_up(w.d1, o);
      if (lane >= o) w = syn_ff_compose(u, w);
    }
    syn_ff ex;
    ex.d0 = __shfl_up(w.d0, 1);
    ex.d1 = __shfl_up(w.d1, 1);
    if (lane == 0) ex = syn_ff{0, 0};
    if (lane < NW) wtot[NW + lane] = ex;
    if (lane == NW - 1) wtot[2 * NW] = w;
  }
  __syncthreads();
  const syn_ff pre = wtot[NW + wv];
  syn_ff prev;
  prev.d0 = __shfl_up(incl.d0, 1);
  prev.d1 = __shfl_up(incl.d1, 1);
  if (lane == 0) prev = syn_ff{0, 0};
  *total = wtot[2 * NW];
  return syn_ff_compose(pre, prev);
}

__device__ __forceinline__ int syn_exponent(double v) {
  return (int)(((unsigned long long)__double_as_longlong(v) >> 52) & 0x7ff) - 1023;
}

#define SYN_TL_PER_THREAD (SYN_PH_TILE / KWY_THREADS)  // 16 samples per thread in the parallel phases
#define SYN_SLOW (-100000)

// phase B1: plain f64 sum of every tile's increments (an estimate of where each tile starts)
__device__ __forceinline__ void syn_tile_sums_body(const double *__restrict__ inc, int64_t y_length,
                                                              double *__restrict__ tsum) {
  __shared__ double red[8];
  const int64_t tile0 = (int64_t)blockIdx.x * SYN_PH_TILE;
  double s = 0.0;
  for (int i = threadIdx.x; i < SYN_PH_TILE; i += KWY_THREADS)
    if (tile0 + i < y_length) s += inc[tile0 + i];
  s = kwy_block_sum(s, red);
  if (threadIdx.x == 0) tsum[blockIdx.x] = s;
}

// phase B2: per tile, guess the binade from the estimated start/end phase; if the tile safely stays
// inside one binade, reduce its samples to ONE parity map (summary); otherwise mark it slow.  A slow tile whose
// start is estimated in binade ka gets the maps of its 16 chunks of 256 samples for BOTH ka and ka + 1: the chain
// then steps over the chunks and looks at single samples only inside the chunk where the phase changes binade.
#define SYN_CHUNK 256
#define SYN_NCHUNK (SYN_PH_TILE / SYN_CHUNK)
struct syn_chunks { int ka; int pad; syn_ff m[SYN_NCHUNK][2]; };      // ka == SYN_SLOW: no chunk maps
__device__ __forceinline__ void syn_tile_summary_body(const double *__restrict__ inc,
                                                                 int64_t y_length,
                                                                 const double *__restrict__ tsum, int ntiles,
                                                                 long long *__restrict__ summ /* 3 per tile */,
                                                                 syn_chunks *__restrict__ chunks) {
  __shared__ syn_ff wtot[2 * KWY_WAVES + 1];
  __shared__ double s_lo;
  const int t = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    double lo = 0.0;
    for (int i = 0; i < t; ++i) lo += tsum[i];
    s_lo = lo;
  }
  __syncthreads();
  const double lo = s_lo, hi = lo + tsum[t];
  int k = SYN_SLOW;
  if (lo > 0.0) {
    const int ka = syn_exponent(lo * (1.0 - 1e-9)), kb = syn_exponent(hi * (1.0 + 1e-9));
    if (ka == kb) k = ka;
  }
  long long *o = summ + 3 * (int64_t)t;
  const int64_t tile0 = (int64_t)t * SYN_PH_TILE;
  if (k == SYN_SLOW) {
    if (tid == 0) { o[0] = 0; o[1] = 0; o[2] = SYN_SLOW; }
    const int ka = lo > 0.0 ? syn_exponent(lo * (1.0 - 1e-9)) : SYN_SLOW;
    if (tid == 0) chunks[t].ka = ka;
    if (ka == SYN_SLOW) return;
    // 16 threads x 16 samples = one chunk: the maps of a row of 16 lanes composed in order
    syn_ff m0 = {0, 0}, m1 = {0, 0};
#pragma unroll
    for (int j = 0; j < SYN_TL_PER_THREAD; ++j) {
      const int64_t n = tile0 + tid * SYN_TL_PER_THREAD + j;
      if (n < y_length) {
        const double cj = inc[n];
        m0 = syn_ff_compose(m0, syn_ff_make(cj, ka));
        m1 = syn_ff_compose(m1, syn_ff_make(cj, ka + 1));
      }
    }
#pragma unroll
    for (int ofs = 1; ofs < 16; ofs <<= 1) {
      syn_ff u0, u1;
      u0.d0 = __shfl_up(m0.d0, ofs, 16); u0.d1 = __shfl_up(m0.d1, ofs, 16);
      u1.d0 = __shfl_up(m1.d0, ofs, 16); u1.d1 = __shfl_up(m1.d1, ofs, 16);
      if ((tid & 15) >= ofs) { m0 = syn_ff_compose(u0, m0); m1 = syn_ff_compose(u1, m1); }
    }
    if ((tid & 15) == 15) { chunks[t].m[tid >> 4][0] = m0; chunks[t].m[tid >> 4][1] = m1; }
    return;
  }
  if (tid == 0) chunks[t].ka = SYN_SLOW;
  syn_ff mine = {0, 0};
#pragma unroll
  for (int j = 0; j < SYN_TL_PER_THREAD; ++j) {
    int64_t n = tile0 + tid * SYN_TL_PER_THREAD + j;
    if (n < y_length) mine = syn_ff_compose(mine, syn_ff_make(inc[n], k));
  }
  syn_ff all;
  (void)syn_ff_block_exscan<KWY_WAVES>(mine, wtot, &all);
  if (tid == 0) { o[0] = all.d0; o[1] = all.d1; o[2] = k; }
}

// phase B3: the serial chain over tiles.  Fast tiles cost a few scalar operations (apply the
// summary to the exact running phase).  Slow tiles (binade changes, the very beginning) are cut
// here, with workgroup-wide scans, into segments that stay inside one binade: where each starts,
// where it ends and the exact phase it starts from; their samples' phases (an fmod each: most of
// what a round of this one workgroup used to cost) are left to the parallel pass like the fast
// tiles'.  Only the samples that carry the phase into the next binade are produced here.
#define SYN_MAXSEG 30
#define SYN_WARM 512              // additions made one by one when the phase leaves zero
struct syn_segs {                 // of one slow tile
  int n;
  int start[SYN_MAXSEG], end[SYN_MAXSEG];
  double tp[SYN_MAXSEG];
};
__device__ __forceinline__ void syn_phase_body(const double *__restrict__ inc,
                                                             int64_t y_length,
                                                             const long long *__restrict__ summ,
                                                             double *__restrict__ tin,
                                                             syn_segs *__restrict__ segs,
                                                             const syn_chunks *__restrict__ chunks,
                                                             double *__restrict__ wrap, long long *__restrict__ dbg) {
  __shared__ syn_ff wtot[2 * (SYN_PH_THREADS / 64) + 1];
  __shared__ int s_cross, s_tix, s_pos, s_nseg;
  __shared__ double s_tp;
  __shared__ double s_c[SYN_WARM + 1], s_tps[SYN_WARM];
  const int tid = threadIdx.x;
  const int ntiles = (int)((y_length + SYN_PH_TILE - 1) / SYN_PH_TILE);
  double tp = 0.0;  // running phase (uniform across the block)
  int tix = 0;
  // in-kernel stamps (diagnostic: tools/syn_phase_stamps.py): [40] clock64 of the launch, [41] the same on the 100 MHz
  // wall clock, [42] spent on fast tiles, [43] on slow tiles, [44] fast tiles, [45] slow tiles, [46] rounds
  long long t_fast = 0, t_slow = 0, t_mark = dbg ? clock64() : 0;
  const long long t_begin = t_mark, w_begin = dbg ? wall_clock64() : 0;
  int n_fast = 0, n_slow = 0, n_rounds = 0;
#define SYN_DBG_FAST() do { if (dbg) { const long long t_ = clock64(); t_fast += t_ - t_mark; t_mark = t_; ++n_fast; } } while (0)
#define SYN_DBG_SLOW() do { if (dbg) { const long long t_ = clock64(); t_slow += t_ - t_mark; t_mark = t_; ++n_slow; } } while (0)
  while (true) {
    // ---- fast tiles: the first wavefront alone takes them one after the other (sixteen wavefronts stepping through
    //      the same scalar chain took sixteen times the issue slots), until a tile needs the whole workgroup
    if (tid < 64) {
      // (the next tile's summary is fetched while this one is applied: the chain waits for the phase, not for memory)
      long long d0n = 0, d1n = 0, kn = SYN_SLOW;
      if (tix < ntiles) { d0n = summ[3 * tix]; d1n = summ[3 * tix + 1]; kn = summ[3 * tix + 2]; }
      while (tix < ntiles) {
        const long long d0 = d0n, d1 = d1n;
        const int k = (int)kn;
        if (tix + 1 < ntiles) { d0n = summ[3 * (tix + 1)]; d1n = summ[3 * (tix + 1) + 1]; kn = summ[3 * (tix + 1) + 2]; }
        if (!(k != SYN_SLOW && tp > 0.0 && syn_exponent(tp) == k)) break;
        const unsigned long long tb = (unsigned long long)__double_as_longlong(tp);
        const long long m_in = (long long)((tb & 0xfffffffffffffULL) | 0x10000000000000ULL);
        const long long m_out = m_in + ((m_in & 1) ? d1 : d0);
        if ((m_out >> 53) != 0) break;
        if (tid == 0) tin[tix] = tp;
        tp = ldexp((double)m_out, k - 52);
        ++tix;
        SYN_DBG_FAST();
      }
      if (tid == 0) { s_tix = tix; s_tp = tp; }
    }
    __syncthreads();
    tix = s_tix;
    tp = s_tp;
    __syncthreads();
    if (tix >= ntiles) break;
    const int64_t tile0 = (int64_t)tix * SYN_PH_TILE;
    const int tile_n = (int)min((int64_t)SYN_PH_TILE, y_length - tile0);
    if (tid == 0) tin[tix] = -1.0;  // cut into segments here
    int nseg = 0;
    int pos = 0;  // first element of the tile not yet produced
    // ---- a tile with chunk maps: the first wavefront steps over the chunks with the map of the current binade and
    //      goes down to single samples (four per lane, wave-level scans, no barrier) only in a chunk where the
    //      phase leaves the binade.  What it cannot finish (no maps for the binade reached) is left to the rounds.
    if (tid < 64 && tp > 0.0 && chunks[tix].ka != SYN_SLOW) {
      const int lane = tid;
      const int ka = chunks[tix].ka;
      // (the 32 maps of the tile in one load, handed out by lane: a load per chunk would be a round trip per chunk)
      const syn_ff mymap = lane < 2 * SYN_NCHUNK ? chunks[tix].m[lane >> 1][lane & 1] : syn_ff{0, 0};
      int seg_start = 0;
      double seg_tp = tp;
      bool give_up = false;
      for (int ch = 0; ch < SYN_NCHUNK && pos < tile_n && !give_up; ++ch) {
        int k = syn_exponent(tp);
        if (k < ka || k > ka + 1) break;
        {
          syn_ff cm;
          cm.d0 = __shfl(mymap.d0, 2 * ch + (k - ka));
          cm.d1 = __shfl(mymap.d1, 2 * ch + (k - ka));
          const unsigned long long tb = (unsigned long long)__double_as_longlong(tp);
          const long long m_in = (long long)((tb & 0xfffffffffffffULL) | 0x10000000000000ULL);
          const long long m_out = m_in + ((m_in & 1) ? cm.d1 : cm.d0);
          if ((m_out >> 53) == 0) {                    // the whole chunk stays in the binade
            tp = ldexp((double)m_out, k - 52);
            pos = min(tile_n, (ch + 1) * SYN_CHUNK);
            continue;
          }
        }
        // the chunk's samples one by one: lane l has samples 4 l .. 4 l + 3 of the chunk
        double cc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = ch * SYN_CHUNK + 4 * lane + j;
          cc[j] = i < tile_n ? inc[tile0 + i] : 0.0;
        }
        int p0 = ch * SYN_CHUNK;                       // first sample of the chunk not yet passed
        const int pend = min(tile_n, (ch + 1) * SYN_CHUNK);
        while (p0 < pend) {
          k = syn_exponent(tp);
          if (k < ka || k > ka + 1) { give_up = true; break; }
          const unsigned long long tb = (unsigned long long)__double_as_longlong(tp);
          const long long m_in = (long long)((tb & 0xfffffffffffffULL) | 0x10000000000000ULL);
          syn_ff f[4], mine = {0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = ch * SYN_CHUNK + 4 * lane + j;
            f[j] = (i >= p0 && i < pend) ? syn_ff_make(cc[j], k) : syn_ff{0, 0};
            mine = syn_ff_compose(mine, f[j]);
          }
          syn_ff incl = mine;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            syn_ff u;
            u.d0 = __shfl_up(incl.d0, o);
            u.d1 = __shfl_up(incl.d1, o);
            if (lane >= o) incl = syn_ff_compose(u, incl);
          }
          syn_ff ex;
          ex.d0 = __shfl_up(incl.d0, 1);
          ex.d1 = __shfl_up(incl.d1, 1);
          if (lane == 0) ex = syn_ff{0, 0};
          long long m = m_in + ((m_in & 1) ? ex.d1 : ex.d0);
          long long mv[4];
          int my_cross = pend;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = ch * SYN_CHUNK + 4 * lane + j;
            if (i >= p0 && i < pend) {
              m += (m & 1) ? f[j].d1 : f[j].d0;
              if ((m >> 53) != 0 && my_cross == pend) my_cross = i;
            }
            mv[j] = m;
          }
          int cross = my_cross;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) cross = min(cross, __shfl_xor(cross, o));
          // the phase in front of `cross` (or at the chunk's end): the mantissa of sample cross - 1
          if (cross > p0) {
            const int q = cross - 1 - ch * SYN_CHUNK;                 // its place in the chunk
            const long long cand = (q & 3) == 0 ? mv[0] : (q & 3) == 1 ? mv[1] : (q & 3) == 2 ? mv[2] : mv[3];
            const long long mq = __shfl(cand, q >> 2);
            tp = ldexp((double)mq, k - 52);
          }
          if (cross >= pend) { p0 = pend; break; }
          // sample `cross` carries the phase into the next binade: close the segment in front of it, add for real
          if (cross > seg_start) {
            if (nseg < SYN_MAXSEG) {
              if (lane == 0) { segs[tix].start[nseg] = seg_start; segs[tix].end[nseg] = cross; segs[tix].tp[nseg] = seg_tp; }
              ++nseg;
            } else { give_up = true; break; }          // (the rounds below produce the rest themselves)
          }
          {
            const int q = cross - ch * SYN_CHUNK;
            const double cand = (q & 3) == 0 ? cc[0] : (q & 3) == 1 ? cc[1] : (q & 3) == 2 ? cc[2] : cc[3];
            tp = tp + __shfl(cand, q >> 2);
          }
          if (lane == 0) wrap[tile0 + cross] = fmod(tp, SYN_TWO_PI);
          p0 = cross + 1;
          seg_start = p0;
          seg_tp = tp;
        }
        if (give_up) { pos = seg_start; tp = seg_tp; break; }
        pos = pend;
      }
      // the segment that is still open
      if (!give_up && pos > seg_start) {
        if (nseg < SYN_MAXSEG) {
          if (lane == 0) { segs[tix].start[nseg] = seg_start; segs[tix].end[nseg] = pos; segs[tix].tp[nseg] = seg_tp; }
          ++nseg;
        } else { pos = seg_start; tp = seg_tp; }
      }
      if (lane == 0) { s_pos = pos; s_nseg = nseg; s_tp = tp; }
    } else if (tid == 0) {
      s_pos = 0; s_nseg = 0; s_tp = tp;
    }
    __syncthreads();
    pos = s_pos;
    nseg = s_nseg;
    tp = s_tp;
    __syncthreads();
    double c[SYN_PH_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SYN_PH_PER_THREAD; ++j) {
      int i = tid * SYN_PH_PER_THREAD + j;
      c[j] = (i < tile_n && pos < tile_n) ? inc[tile0 + i] : 0.0;
    }
    while (pos < tile_n) {
      if (tid == 0) s_cross = tile_n;
      __syncthreads();

      int cross;
      if (tp == 0.0) {
        // 0 + c is a plain addition, and while the increments are zero (an unvoiced beginning: leading silence
        // is the normal case for recorded speech) nothing moves: skip to the first non-zero increment at once
        // instead of taking the samples one per round
        int my_first = tile_n;
#pragma unroll
        for (int j = SYN_PH_PER_THREAD - 1; j >= 0; --j) {
          const int i = tid * SYN_PH_PER_THREAD + j;
          if (i >= pos && i < tile_n && c[j] != 0.0) my_first = i;
        }
        {
          int wmin = my_first;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o));
          if ((tid & 63) == 0 && wmin < tile_n) atomicMin(&s_cross, wmin);
        }
        __syncthreads();
        cross = s_cross;
#pragma unroll
        for (int j = 0; j < SYN_PH_PER_THREAD; ++j) {
          const int i = tid * SYN_PH_PER_THREAD + j;
          if (i >= pos && i < cross) wrap[tile0 + i] = 0.0;      // fmod(0 + 0, 2 pi)
        }
      } else {
        const int k = syn_exponent(tp);
        const unsigned long long tb = (unsigned long long)__double_as_longlong(tp);
        const long long m_in = (long long)((tb & 0xfffffffffffffULL) | 0x10000000000000ULL);
        syn_ff f[SYN_PH_PER_THREAD];
        syn_ff mine = {0, 0};
#pragma unroll
        for (int j = 0; j < SYN_PH_PER_THREAD; ++j) f[j] = syn_ff{0, 0};
        if (((tid | 63) + 1) * SYN_PH_PER_THREAD > pos) {     // (a wavefront whose samples are all done has no maps to make)
#pragma unroll
          for (int j = 0; j < SYN_PH_PER_THREAD; ++j) {
            int i = tid * SYN_PH_PER_THREAD + j;
            if (i >= pos && i < tile_n) f[j] = syn_ff_make(c[j], k);
            mine = syn_ff_compose(mine, f[j]);
          }
        }
        syn_ff all;
        const syn_ff excl = syn_ff_block_exscan<SYN_PH_THREADS / 64>(mine, wtot, &all);
        long long m = m_in + ((m_in & 1) ? excl.d1 : excl.d0);
        int my_cross = tile_n;
        long long mv[SYN_PH_PER_THREAD];
#pragma unroll
        for (int j = 0; j < SYN_PH_PER_THREAD; ++j) {
          int i = tid * SYN_PH_PER_THREAD + j;
          if (i >= pos && i < tile_n) {
            m += (m & 1) ? f[j].d1 : f[j].d0;
            if ((m >> 53) != 0 && my_cross == tile_n) my_cross = i;
          }
          mv[j] = m;
        }
        // (behind a crossing every mantissa is out of range: one atomic per wavefront, not one per thread)
        {
          int wmin = my_cross;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o));
          if ((tid & 63) == 0 && wmin < tile_n) atomicMin(&s_cross, wmin);
        }
        __syncthreads();
        cross = s_cross;
        const bool direct = nseg >= SYN_MAXSEG;      // (no room for another segment: its samples are produced here)
#pragma unroll
        for (int j = 0; j < SYN_PH_PER_THREAD; ++j) {
          int i = tid * SYN_PH_PER_THREAD + j;
          if (i >= pos && i < cross) {
            if (direct) wrap[tile0 + i] = fmod(ldexp((double)mv[j], k - 52), SYN_TWO_PI);
            if (i == cross - 1) s_tp = ldexp((double)mv[j], k - 52);
          }
        }
        if (cross > pos && !direct) {
          if (tid == 0) { segs[tix].start[nseg] = pos; segs[tix].end[nseg] = cross; segs[tix].tp[nseg] = tp; }
          ++nseg;
        }
        __syncthreads();
        if (cross > pos) tp = s_tp;
      }
      if (cross < tile_n && tp == 0.0) {
        // The phase leaves zero: the next few hundred additions cross a binade every few samples (each crossing a
        // round of the whole workgroup).  One thread makes them for real, one after the other; their fmods are
        // taken by as many threads.
        const int nw = min(SYN_WARM, tile_n - cross);
        if (tid < nw) s_c[tid] = inc[tile0 + cross + tid];
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          int q = 0;
          for (; q + 8 <= nw; q += 8) {          // eight at a time: one LDS round trip per eight dependent additions
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s_c[q + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { t = t + v[u]; v[u] = t; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s_tps[q + u] = v[u];
          }
          for (; q < nw; ++q) { t = t + s_c[q]; s_tps[q] = t; }
          s_tp = t;
        }
        __syncthreads();
        if (tid < nw) wrap[tile0 + cross + tid] = fmod(s_tps[tid], SYN_TWO_PI);
        tp = s_tp;
        pos = cross + nw;
      } else if (cross < tile_n) {
        // this addition leaves the binade: do it for real
        tp = tp + inc[tile0 + cross];
        if (tid == 0) wrap[tile0 + cross] = fmod(tp, SYN_TWO_PI);
        pos = cross + 1;
      } else {
        pos = tile_n;
      }
      __syncthreads();
      ++n_rounds;
    }
    if (tid == 0) segs[tix].n = nseg;
    ++tix;
    SYN_DBG_SLOW();
  }
  if (dbg && tid == 0) {
    dbg[40] = clock64() - t_begin; dbg[41] = wall_clock64() - w_begin; dbg[42] = t_fast; dbg[43] = t_slow;
    dbg[44] = n_fast; dbg[45] = n_slow; dbg[46] = n_rounds;
  }
}

// phase B4, in parallel: every sample's exact phase from the exact phase its tile (or its segment of a slow tile)
// starts from
__device__ __forceinline__ void syn_tile_apply_body(const double *__restrict__ inc, int64_t y_length,
                                                               const double *__restrict__ tin,
                                                               const syn_segs *__restrict__ segs,
                                                               double *__restrict__ wrap) {
  __shared__ syn_ff wtot[2 * KWY_WAVES + 1];
  const int t = blockIdx.x, tid = threadIdx.x;
  const double tp_tile = tin[t];
  const bool fast = tp_tile > 0.0;
  const int ns = fast ? 1 : segs[t].n;
  const int64_t base = (int64_t)t * SYN_PH_TILE + (int64_t)tid * SYN_TL_PER_THREAD;
  double c[SYN_TL_PER_THREAD];
#pragma unroll
  for (int j = 0; j < SYN_TL_PER_THREAD; ++j) c[j] = (base + j < y_length) ? inc[base + j] : 0.0;
  // a sample's exact phase, from the segment it lies in; its fmod is taken once, behind the segments' scans
  double tpv[SYN_TL_PER_THREAD];
#pragma unroll
  for (int j = 0; j < SYN_TL_PER_THREAD; ++j) tpv[j] = -1.0;
  for (int q = 0; q < ns; ++q) {
    const double tp = fast ? tp_tile : segs[t].tp[q];
    const int s0 = fast ? 0 : segs[t].start[q], s1 = fast ? SYN_PH_TILE : segs[t].end[q];   // samples of the tile
    const int k = syn_exponent(tp);
    const unsigned long long tb = (unsigned long long)__double_as_longlong(tp);
    const long long m_in = (long long)((tb & 0xfffffffffffffULL) | 0x10000000000000ULL);
    // (a wavefront none of whose samples lies in the segment only takes part in the scan)
    const bool active = ((tid | 63) + 1) * SYN_TL_PER_THREAD > s0 && (tid & ~63) * SYN_TL_PER_THREAD < s1;
    syn_ff f[SYN_TL_PER_THREAD];
    syn_ff mine = {0, 0};
    if (active) {
#pragma unroll
      for (int j = 0; j < SYN_TL_PER_THREAD; ++j) {
        const int li = tid * SYN_TL_PER_THREAD + j;
        f[j] = (li >= s0 && li < s1 && base + j < y_length) ? syn_ff_make(c[j], k) : syn_ff{0, 0};
        mine = syn_ff_compose(mine, f[j]);
      }
    }
    syn_ff all;
    const syn_ff excl = syn_ff_block_exscan<KWY_WAVES>(mine, wtot, &all);
    if (active) {
      long long m = m_in + ((m_in & 1) ? excl.d1 : excl.d0);
#pragma unroll
      for (int j = 0; j < SYN_TL_PER_THREAD; ++j) {
        const int li = tid * SYN_TL_PER_THREAD + j;
        if (li >= s0 && li < s1 && base + j < y_length) {
          m += (m & 1) ? f[j].d1 : f[j].d0;
          tpv[j] = ldexp((double)m, k - 52);
        }
      }
    }
    __syncthreads();      // (wtot is reused by the next segment)
  }
#pragma unroll
  for (int j = 0; j < SYN_TL_PER_THREAD; ++j)
    if (tpv[j] >= 0.0) wrap[base + j] = fmod(tpv[j], SYN_TWO_PI);
}

__device__ __forceinline__ bool syn_is_pulse(const double *__restrict__ wrap, int64_t n, int64_t y_length) {
  return n < y_length - 1 && fabs(wrap[n + 1] - wrap[n]) > KWY_PI;
}

// phase D: pulses per tile
__device__ __forceinline__ void syn_pulse_count_body(const double *__restrict__ wrap,
                                                                int64_t y_length, int *__restrict__ cnt) {
  __shared__ int sh[KWY_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SYN_TILE + (int64_t)threadIdx.x * SYN_TILE_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < SYN_TILE_PER_THREAD; ++j) c += syn_is_pulse(wrap, base + j, y_length) ? 1 : 0;
  int tot;
  (void)syn_block_exscan_int(c, sh, &tot);
  if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// phase E: ordered pulse list
__device__ __forceinline__ void syn_pulse_emit_body(const double *__restrict__ wrap,
                                                               int64_t y_length, int fs,
                                                               const int *__restrict__ tile_off,
                                                               int cap, int32_t *__restrict__ pidx,
                                                               double *__restrict__ pshift) {
  __shared__ int sh[KWY_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SYN_TILE + (int64_t)threadIdx.x * SYN_TILE_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < SYN_TILE_PER_THREAD; ++j) c += syn_is_pulse(wrap, base + j, y_length) ? 1 : 0;
  int tot;
  int pos = tile_off[blockIdx.x] + syn_block_exscan_int(c, sh, &tot);
#pragma unroll
  for (int j = 0; j < SYN_TILE_PER_THREAD; ++j) {
    int64_t n = base + j;
    if (syn_is_pulse(wrap, n, y_length)) {
      if (pos < cap) {
        double y1 = wrap[n] - SYN_TWO_PI;
        double y2 = wrap[n + 1];
        double xx = -y1 / (y2 - y1);
        pidx[pos] = (int32_t)n;
        pshift[pos] = xx / fs;
      }
      ++pos;
    }
  }
}

// ---- the placement kernels: one launch per step for a batch of utterances (blockIdx.y = utterance) ----------------
// The pulse placement -- everything that depends on f0 only -- and the rendering of the pulses are separate steps, so
// that a pipeline can place the pulses on another stream while the spectral features are still being computed
// (kwy_synth_plan_dev / kwy_synth_render_dev); kwy_synthesize_dev runs both.
struct syn_plan {
  int *npulse;             // 16 ints
  int *tile_cnt;           // nt + 1: pulses per output tile, then their offsets
  unsigned char *vuv8;     // y_length
  int32_t *pidx;           // cap
  double *pshift;          // cap
};
struct syn_plan_view {
  const double *f0;
  syn_params p;
  syn_plan pl;
  double *incr, *wrap, *tsum, *tin;    // scratch: increments, wrapped phase, tile sums, tile start phases
  long long *summ;
  syn_segs *segs;
  syn_chunks *chunks;
  int cap;
};
struct syn_plan_batch {
  int n;
  long long *dbg;
  syn_plan_view u[KWY_BATCH_MAX];
};
__device__ __forceinline__ int syn_npt(int64_t y_length) { return (int)((y_length + SYN_PH_TILE - 1) / SYN_PH_TILE); }
__device__ __forceinline__ int syn_nt(int64_t y_length) { return (int)((y_length + SYN_TILE - 1) / SYN_TILE); }

__global__ __launch_bounds__(KWY_THREADS) void k_syn_inc(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  syn_inc_body(v.f0, v.p, v.incr, v.pl.vuv8);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_tile_sums(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  if ((int)blockIdx.x >= syn_npt(v.p.y_length)) return;
  syn_tile_sums_body(v.incr, v.p.y_length, v.tsum);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_tile_summary(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  const int npt = syn_npt(v.p.y_length);
  if ((int)blockIdx.x >= npt) return;
  syn_tile_summary_body(v.incr, v.p.y_length, v.tsum, npt, v.summ, v.chunks);
}
__global__ __launch_bounds__(SYN_PH_THREADS) void k_syn_phase(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.x];
  syn_phase_body(v.incr, v.p.y_length, v.summ, v.tin, v.segs, v.chunks, v.wrap, b.dbg);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_tile_apply(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  if ((int)blockIdx.x >= syn_npt(v.p.y_length)) return;
  syn_tile_apply_body(v.incr, v.p.y_length, v.tin, v.segs, v.wrap);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_pulse_count(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  if ((int)blockIdx.x >= syn_nt(v.p.y_length)) return;
  syn_pulse_count_body(v.wrap, v.p.y_length, v.pl.tile_cnt);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_scan_counts(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.x];
  syn_scan_counts_body(v.pl.tile_cnt, syn_nt(v.p.y_length), v.pl.npulse);
}
__global__ __launch_bounds__(KWY_THREADS) void k_syn_pulse_emit(syn_plan_batch b) {
  const syn_plan_view &v = b.u[blockIdx.y];
  if ((int)blockIdx.x >= syn_nt(v.p.y_length)) return;
  syn_pulse_emit_body(v.wrap, v.p.y_length, v.p.fs, v.pl.tile_cnt, v.cap, v.pl.pidx, v.pl.pshift);
}

// The pulse kernel runs with as many threads as a radix-8 pass of its transform has butterflies (N/16: 128 at 48 kHz)
// -- every wavefront works in every pass, and the registers of the idle half buy two more workgroups per CU -- but
// keeps the arithmetic of the 256-thread form it replaced, bit for bit: thread t stands for the V = 256/NT "virtual"
// threads t + NT g.  Pair twiddles: exp(-2 pi i k / N) for k = t + NT r as base[r % V] times an 8th root of unity,
// base[g] = exp(-2 pi i (t + NT g) / N); block sums: the partial sums of the virtual threads, wavefront by
// wavefront (syn_block_sum_v).
template <int LOG2N>
struct syn_pulse_nt { static constexpr int value = LOG2N <= 10 ? 64 : (LOG2N == 11 ? 128 : 256); };

template <int LOG2H, int NT, int V>
__device__ __forceinline__ kwy_c syn_pair_twiddle(const kwy_c (&base)[V], int r) {
  constexpr int N = 2 << LOG2H;
  constexpr int OCT = 8 * (NT * V) / N;                     // >= 1: N <= 2048 here, or V = 1 and NT >= N/8
  return kwy_tw_octant(base[r % V], OCT * (r / V));
}

template <int LOG2H, int NT, int V>
__device__ inline void syn_rfft_inplace(kwy_c *z, const kwy_c *__restrict__ tw, const kwy_c (&base)[V],
                                        const kwy_c *__restrict__ twN) {
  constexpr int H = 1 << LOG2H, N = 2 * H;
  constexpr bool TABLE = 8 * (NT * V) / N < 1;              // the long transforms read the pair twiddles
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
      kwy_c w;
      if constexpr (TABLE) w = twN[k]; else w = syn_pair_twiddle<LOG2H, NT, V>(base, r);
      const kwy_c A = z[k], Bc = z[H - k];
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

template <int LOG2H, int NT, int V>
__device__ inline void syn_irfft_inplace(kwy_c *z, const kwy_c *__restrict__ tw, const kwy_c (&base)[V],
                                         const kwy_c *__restrict__ twN) {
  constexpr int H = 1 << LOG2H, N = 2 * H;
  constexpr bool TABLE = 8 * (NT * V) / N < 1;
  const int tid = kwy_tid_opaque();
  __syncthreads();
#pragma unroll
  for (int r = 0; r * NT <= H / 2; ++r) {
    const int k = tid + NT * r;
    if (k > H / 2) continue;
    if (k == 0) {
      const double ar = z[0].x, br = z[H].x;
      z[0] = {ar + br, ar - br};
    } else {
      kwy_c w;
      if constexpr (TABLE) w = twN[k]; else w = syn_pair_twiddle<LOG2H, NT, V>(base, r);
      const kwy_c A = z[k], Bc = z[H - k];
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

// sum over the workgroup of the V partial sums every thread carries, in the order of a 256-thread workgroup whose
// thread t + NT g carries part[g]: wavefront sums, then the four of them left to right.  red: >= 4 doubles.
template <int NT, int V>
__device__ __forceinline__ double syn_block_sum_v(const double (&part)[V], double *red) {
  double ws[V];
#pragma unroll
  for (int g = 0; g < V; ++g) ws[g] = kwy_wave_sum(part[g]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int g = 0; g < V; ++g) red[(threadIdx.x >> 6) + (NT / 64) * g] = ws[g];
  }
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT * V / 64; ++i) s += red[i];
  return s;
}

__device__ __forceinline__ double syn_safe_ap(double x) {
  return fmax(0.001, fmin(0.999999999999, x));
}

// One pulse per workgroup iteration; its response goes to slot (pulse - first_pulse) of `resp`.  LDS: one FFT buffer
// (in-place transforms), the half log-spectrum of the aperiodic part (formed with the periodic one's in the single pass
// over the interpolated envelope / aperiodicity rows, which are not kept), a small twiddle table -- 27 KB at 48 kHz:
// five workgroups of two wavefronts per CU, all ten at work in every pass (the 256-thread form of rounds 1-4 held
// 51 KB and kept half of its twelve wavefronts per CU idle through the seven transforms of a pulse).  What is kept
// of the periodic response (H samples and a scalar) takes the place of the aperiodic log-spectrum when that moves
// into the FFT buffer; the minimum-phase spectrum waits in registers while the noise is transformed.
// One utterance of a rendering launch (descriptors by value in the kernel arguments, as for the analysis kernels):
// a launch renders the pulses of up to KWY_BATCH_MAX utterances, each with its own slice of the grid.
struct syn_view {
  const double *sp, *ap;
  syn_params p;
  const int32_t *pidx;
  const double *pshift;
  const unsigned char *vuv8;
  const int *npulse, *tile_off;
  int cap, slots, nt;
  double *resp, *y;
};
typedef kwy_batch<syn_view> syn_batch;

template <int LOG2N, bool DIRECT>
__global__ __launch_bounds__(syn_pulse_nt<LOG2N>::value, LOG2N <= 11 ? 3 : (LOG2N == 12 ? 2 : 1)) void k_syn_pulse(
    syn_batch batch, kwy_randn_src rs, const uint4 *__restrict__ poly,
    const kwy_c *__restrict__ twH, const kwy_c *__restrict__ twN,
    const double *__restrict__ dc_remover, long long *__restrict__ dbg) {
  constexpr int NT = syn_pulse_nt<LOG2N>::value, V = KWY_THREADS / NT;
  // diagnostic stamps (tools/syn_pulse_stamps.py): the first voiced pulse that a workgroup from dbg[63] on reaches
  bool stamping = false;
#define SYN_STAMP(n) do { if (stamping && threadIdx.x == 0) dbg[n] = clock64(); } while (0)
  const int utt = batch.find(blockIdx.x);
  const int lblock = (int)blockIdx.x - batch.start[utt], lgrid = batch.start[utt + 1] - batch.start[utt];
  const double *__restrict__ sp = batch.u[utt].sp, *__restrict__ ap = batch.u[utt].ap;
  const syn_params p = batch.u[utt].p;
  const int32_t *__restrict__ pidx = batch.u[utt].pidx;
  const double *__restrict__ pshift = batch.u[utt].pshift;
  const unsigned char *__restrict__ vuv8 = batch.u[utt].vuv8;
  const int *__restrict__ npulse = batch.u[utt].npulse;
  const int cap = batch.u[utt].cap;
  // the slots' round (every pulse its own slot), or the pulses beyond them (DIRECT: one workgroup, in order, onto y)
  const int first_pulse = DIRECT ? batch.u[utt].slots : 0, slots = DIRECT ? cap : batch.u[utt].slots;
  double *__restrict__ resp = batch.u[utt].resp;
  double *__restrict__ y = batch.u[utt].y;
  constexpr int N = 1 << LOG2N, H = N / 2, K = H + 1;
  constexpr int C = N / NT;                // draws / output samples per thread: sample tid + NT m
  constexpr int RK = (H + 1 + NT - 1) / NT;
  constexpr int TWL = (H / 8 > 1) ? H / 8 : 1;
  extern __shared__ double smem[];
  double *red = smem;                      // 8
  uint32_t *e = (uint32_t *)(red + 8);     // KWY_EBASE_WORDS
  kwy_c *twl = (kwy_c *)(e + KWY_EBASE_WORDS);  // exp(-2 pi i k / H), k < H/8
  kwy_c *buf = twl + TWL;                  // H+1 complex
  double *lap = (double *)(buf + (H + 1)); // K: half log-spectrum of the aperiodic part

  const int tid = threadIdx.x;
  for (int i = tid; i < TWL; i += NT) twl[i] = twH[i];
  kwy_c twb[V];
#pragma unroll
  for (int g = 0; g < V; ++g) twb[g] = twN[tid + NT * g];
  const int P = min(npulse[0], cap);
  const int pend = min(P, first_pulse + slots);    // this round's pulses
  for (int pp = first_pulse + lblock; pp < pend; pp += lgrid) {
    const int tid = kwy_tid_opaque();  // keeps address arithmetic local to the pulse (no spills across the FFTs)
    const int idx = pidx[pp];
    const int nxt = pidx[min(P - 1, pp + 1)];
    const int noise_size = nxt - idx;
    __syncthreads();
    if (noise_size <= 0) continue;  // last pulse: zero response
    const int ns_used = min(noise_size, N);
    const double current_vuv = vuv8[idx] ? 1.0 : 0.0;
    const double current_time = idx / (double)p.fs;
    const double shift = pshift[pp];

    // ---- spectral envelope / aperiodic ratio at the pulse time -> the two half log-spectra
    int fl = (int)floor(current_time / p.frame_period);
    int ce = (int)ceil(current_time / p.frame_period);
    if (fl > p.T - 1) fl = (int)p.T - 1;
    if (ce > p.T - 1) ce = (int)p.T - 1;
    const double interpolation = current_time / p.frame_period - fl;
    const double *s0 = sp + (int64_t)fl * K, *s1 = sp + (int64_t)ce * K;
    const double *a0 = ap + (int64_t)fl * K, *a1 = ap + (int64_t)ce * K;
    double rv0 = syn_safe_ap(a0[0]);
    if (fl != ce) rv0 = (1.0 - interpolation) * rv0 + interpolation * syn_safe_ap(a1[0]);
    const bool has_periodic = current_vuv > 0.5 && !(rv0 * rv0 > 0.999);
    stamping = dbg && threadIdx.x == 0 && has_periodic && blockIdx.x >= (unsigned)dbg[63] && dbg[62] == 0 &&
               atomicCAS((unsigned long long *)&dbg[62], 0ull, 1ull) == 0ull;
    SYN_STAMP(0);
    {
      double *L = (double *)buf;
      // (the rows first, all loads of a thread in flight together, then the logarithms)
      double evs[RK], rts[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        const int k = min(tid + NT * r, H);
        double ev, rv;
        if (fl == ce) {
          ev = fabs(s0[k] * p.sp_mul);
          rv = syn_safe_ap(a0[k]);
        } else {
          ev = (1.0 - interpolation) * fabs(s0[k] * p.sp_mul) + interpolation * fabs(s1[k] * p.sp_mul);
          rv = (1.0 - interpolation) * syn_safe_ap(a0[k]) + interpolation * syn_safe_ap(a1[k]);
        }
        evs[r] = ev;
        rts[r] = rv * rv;
      }
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        const int k = tid + NT * r;
        if (k <= H) {
          lap[k] = kwy_log((current_vuv != 0.0) ? evs[r] * rts[r] : evs[r]) / 2.0;
          if (has_periodic) L[k] = kwy_log(evs[r] * (1.0 - rts[r]) + SYN_SAFE) / 2.0;
        }
      }
    }

    // ---- the two responses, periodic (part 0, voiced pulses only) then aperiodic (part 1), through ONE copy of the
    // code: a pulse is seven transforms and the element-wise steps between them -- written out one after the other
    // that was 68 KB of instructions, more than the 64 KB instruction cache two CUs share, run through once per
    // pulse by workgroups that are each somewhere else in it.  As loops (one forward, one inverse transform in the
    // binary) it is a third of that.
    //   part: [mirror -> rfft] [causal fold -> rfft -> exp] ([noise -> rfft]) -> product -> irfft
    // Of the periodic response dc_component and the H samples w[0 .. H) of the inverse transform are needed later
    // (sample i is -dc * dc_remover[i] below H, w[i - H] - dc * dc_remover[i] from H on): they change places with the
    // aperiodic half log-spectrum -- w to `lap`, the log-spectrum into the FFT buffer.
    double dc_component = 0.0;
    SYN_STAMP(1);
#pragma nounroll
    for (int part = has_periodic ? 0 : 1; part < 2; ++part) {
      const int tid = kwy_tid_opaque();
      double *L = (double *)buf;
      if (part == 1) {
        double la[RK], wv[RK];
#pragma unroll
        for (int r = 0; r < RK; ++r) {
          const int k = tid + NT * r;
          la[r] = k <= H ? lap[k] : 0.0;
          wv[r] = (has_periodic && k < H) ? L[k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RK; ++r) {
          const int k = tid + NT * r;
          if (k <= H) { L[k] = la[r]; lap[k] = wv[r]; }
        }
      }
      // minimum-phase spectrum (common.cpp GetMinimumPhaseSpectrum) of the half log-spectrum in the buffer, then, for
      // the aperiodic part, the spectrum of the pulse's noise while the minimum-phase one waits in registers
      kwy_c mp[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) mp[r] = {0.0, 0.0};
#pragma nounroll
      for (int it = 0; it < 2 + part; ++it) {
        const int tid = kwy_tid_opaque();
        if (it == 0) {
          __syncthreads();
          for (int i = H + 1 + tid; i < N; i += NT) L[i] = L[N - i];
          __syncthreads();
        } else if (it == 1) {
          // causal cepstrum: c[0], 2 c[1..H-1], c[H], zeros -- packed reals over the complex bins they came from
          double cv[RK];
#pragma unroll
          for (int r = 0; r < RK; ++r) {
            const int n = tid + NT * r;
            double v = 0.0;
            if (n <= H) {
              v = buf[n].x;
              if (n >= 1 && n < H) v *= 2.0;
            }
            cv[r] = v;
          }
          __syncthreads();
#pragma unroll
          for (int r = 0; r < RK; ++r) {
            const int n = tid + NT * r;
            if (n <= H) L[n] = cv[r];
          }
          for (int n = H + 1 + tid; n < N; n += NT) L[n] = 0.0;
          __syncthreads();
        } else {
#pragma unroll
          for (int r = 0; r < RK; ++r) {
            const int k = tid + NT * r;
            mp[r] = k <= H ? buf[k] : kwy_c{0.0, 0.0};
          }
          __syncthreads();
          // the pulse's noise: draws [dpos, dpos + noise_size) of the stream, dpos = idx - idx[0] (the serial code
          // draws noise_size numbers per pulse); sample d = tid + NT j takes draw d -- from the table, or beyond it
          // from the generator (jump table of the pulse's stream position in the currently idle FFT buffer, thread t
          // makes the C consecutive draws from C t on, which travel through LDS)
          const uint64_t dpos = (uint64_t)(idx - pidx[0]);
          uint32_t raw[C];
          if (dpos + (uint64_t)ns_used <= rs.n) {
#pragma unroll
            for (int j = 0; j < C; ++j) raw[j] = (tid + NT * j < ns_used) ? rs.tab[dpos + tid + NT * j] : 0u;
          } else {
            kwy_rng_block_ebase(dpos, rs.pow2, e);
            kwy_rng rng;
            if constexpr (sizeof(kwy_c) * (H + 1) >= 8192) {
              kwy_rng_build_table<NT>(e, (uint4 *)buf);
              __syncthreads();
              rng = kwy_rng_combine_table((const uint4 *)buf, poly[tid]);
              __syncthreads();
            } else {
              rng = kwy_rng_combine(e, poly[tid]);
            }
            uint32_t *D = (uint32_t *)buf;
#pragma unroll
            for (int j = 0; j < C; ++j) D[C * tid + j] = kwy_rng_randn_raw(rng);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < C; ++j) raw[j] = D[tid + NT * j];
            __syncthreads();
          }
          double nv[C];
          double sum[V];
#pragma unroll
          for (int g = 0; g < V; ++g) sum[g] = 0.0;
#pragma unroll
          for (int j = 0; j < C; ++j) {
            const int d = tid + NT * j;
            nv[j] = kwy_randn_from_raw(raw[j]);
            if (d < ns_used) sum[j % V] += nv[j];             // virtual thread tid + NT (j % V), its element j / V
          }
          const double average = syn_block_sum_v<NT, V>(sum, red) / noise_size;
#pragma unroll
          for (int j = 0; j < C; ++j) {
            const int d = tid + NT * j;
            L[d] = (d < ns_used) ? nv[j] - average : 0.0;
          }
          __syncthreads();
        }
        syn_rfft_inplace<LOG2N - 1, NT, V>(buf, twl, twb, twN);
        if (it == 1) {
          // (three bins a trip: their exp / sincos chains are independent and fill each other's issue gaps)
          constexpr int G = 3;
#pragma nounroll
          for (int k0 = tid; k0 <= H; k0 += G * NT) {
            kwy_c R[G];
#pragma unroll
            for (int g = 0; g < G; ++g) R[g] = buf[min(k0 + g * NT, H)];
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const double tmp = exp(R[g].x / N);
              const double ph = R[g].y / N;
              double sn, cs;
              kwy_sincos_medium(ph, &sn, &cs);   // (a minimum-phase angle: a few pi at most)
              R[g] = {tmp * cs, tmp * sn};
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
              if (k0 + g * NT <= H) buf[k0 + g * NT] = R[g];
          }
          __syncthreads();
        }
      }
      SYN_STAMP(2 + 5 * part);
      if (part == 0) {
        const double coefficient = 2.0 * KWY_PI * shift * p.fs / N;
        for (int k = tid; k <= H; k += NT) {
          const double re = buf[k].x, im = buf[k].y;
          const double re2 = kwy_cos_pi_range(coefficient * k);   // shift * fs < 1: the argument stays in [0, pi]
          const double im2 = sqrt(1.0 - re2 * re2);
          buf[k] = {__builtin_fma(re, re2, im * im2), __builtin_fma(im, re2, -(re * im2))};
        }
      } else {
#pragma unroll
        for (int r = 0; r < RK; ++r) {
          const int k = tid + NT * r;
          if (k <= H) buf[k] = cmulf(mp[r], buf[k]);
        }
      }
      SYN_STAMP(3 + 5 * part);
      syn_irfft_inplace<LOG2N - 1, NT, V>(buf, twl, twb, twN);
      SYN_STAMP(4 + 5 * part);
      if (part == 0) {
        const double *w = (const double *)buf;
        double ps[V];
#pragma unroll
        for (int g = 0; g < V; ++g) {
          ps[g] = 0.0;
          for (int i = tid + NT * g; i < H; i += KWY_THREADS) ps[g] += w[i];
        }
        dc_component = syn_block_sum_v<NT, V>(ps, red);
        SYN_STAMP(5);
      }
    }
    {
      const int tid = kwy_tid_opaque();
      SYN_STAMP(10);
      const double *w = (const double *)buf;
      const double sqrt_noise_size = sqrt((double)noise_size);
      // sample i lands at idx - H + 1 + i: into this pulse's slot, or (DIRECT: the one workgroup that takes the
      // pulses beyond the slots, in order) straight onto y
      double *slot = DIRECT ? nullptr : resp + (size_t)(pp - first_pulse) * N;
      const int64_t offset = (int64_t)idx - H + 1;
#pragma unroll
      for (int m = 0; m < C; ++m) {
        int i = tid + NT * m;
        double aper = (i < H) ? w[i + H] : w[i - H];
        double pv = 0.0;
        if (has_periodic) pv = (i < H) ? -dc_component * dc_remover[i] : lap[i - H] - dc_component * dc_remover[i];
        const double r = (pv * sqrt_noise_size + aper) / N;
        if constexpr (DIRECT) {
          const int64_t n = offset + i;
          if (n >= 0 && n < p.y_length) y[n] += r;
        } else {
          slot[i] = r;
        }
      }
      SYN_STAMP(11);
    }
  }
#undef SYN_STAMP
}


// y[n] (+)= the responses of this round's pulses that reach sample n, in pulse order.  The pulses of the
// SYN_TILE-sample tiles within `reach` tiles of the output tile are candidates (tile_off: first pulse of a tile).
__global__ __launch_bounds__(KWY_THREADS) void k_syn_ola(syn_batch batch, int N) {
  const int utt = batch.find(blockIdx.x);
  const double *__restrict__ resp = batch.u[utt].resp;
  const int32_t *__restrict__ pidx = batch.u[utt].pidx;
  const int *__restrict__ tile_off = batch.u[utt].tile_off;
  const int *__restrict__ npulse = batch.u[utt].npulse;
  const int cap = batch.u[utt].cap, nt = batch.u[utt].nt, first_pulse = 0, slots = batch.u[utt].slots;
  const int64_t y_length = batch.u[utt].p.y_length;
  double *__restrict__ y = batch.u[utt].y;
  const int H = N / 2;
  const int tile = (int)blockIdx.x - batch.start[utt];
  const int P = min(npulse[0], cap);
  if (first_pulse >= P && first_pulse > 0) return;     // a round without pulses leaves y alone
  const int reach = (H + SYN_TILE - 1) / SYN_TILE;
  const int ja = max(0, tile - reach), jb = min(nt - 1, tile + reach);
  int p0 = min(tile_off[ja], P), p1 = jb + 1 < nt ? min(tile_off[jb + 1], P) : P;
  p0 = max(p0, first_pulse);
  p1 = min(p1, first_pulse + slots);
  const int64_t n0 = (int64_t)tile * SYN_TILE + threadIdx.x;
  double v[SYN_TILE_PER_THREAD];
#pragma unroll
  for (int m = 0; m < SYN_TILE_PER_THREAD; ++m) {
    const int64_t n = n0 + KWY_THREADS * m;
    v[m] = (first_pulse > 0 && n < y_length) ? y[n] : 0.0;
  }
  for (int pp = p0; pp < p1; ++pp) {
    const int idx = pidx[pp];
    if (pidx[min(P - 1, pp + 1)] - idx <= 0) continue;      // the last pulse has no response
    const double *slot = resp + (size_t)(pp - first_pulse) * N;
    const int64_t base = (int64_t)idx - H + 1;
#pragma unroll
    for (int m = 0; m < SYN_TILE_PER_THREAD; ++m) {
      const int64_t q = n0 + KWY_THREADS * m - base;
      if (q >= 0 && q < N) v[m] += slot[q];
    }
  }
#pragma unroll
  for (int m = 0; m < SYN_TILE_PER_THREAD; ++m) {
    const int64_t n = n0 + KWY_THREADS * m;
    if (n < y_length) y[n] = v[m];
  }
}

// ------------------------------------------------------------------ host side
static int get_dc_remover(kwy_ctx *ctx, int fft_size, const double **out) {
  std::string key = "dcrem:" + std::to_string(fft_size);
  auto it = ctx->d_mats.find(key);
  if (it == ctx->d_mats.end()) {
    std::vector<double> h(fft_size);
    double dc_component = 0.0;
    for (int i = 0; i < fft_size / 2; ++i) {
      h[i] = 0.5 - 0.5 * cos(2.0 * KWY_PI * (i + 1.0) / (1.0 + fft_size));
      h[fft_size - i - 1] = h[i];
      dc_component += h[i] * 2.0;
    }
    for (int i = 0; i < fft_size / 2; ++i) {
      h[i] /= dc_component;
      h[fft_size - i - 1] = h[i];
    }
    double *d = nullptr;
    KWY_HIP(hipMalloc((void **)&d, sizeof(double) * fft_size));
    KWY_HIP(hipMemcpy(d, h.data(), sizeof(double) * fft_size, hipMemcpyHostToDevice));
    it = ctx->d_mats.emplace(key, d).first;
  }
  *out = it->second;
  return KWY_OK;
}

template <int LOG2N>
static int launch_pulse(kwy_ctx *ctx, syn_batch &batch) {
  constexpr int N = 1 << LOG2N, H = N / 2, K = H + 1;
  const kwy_c *twH, *twN;
  const uint4 *poly;
  const double *dcrem;
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N - 1, &twH));
  KWY_TRY(kwy_get_twiddles(ctx, LOG2N, &twN));
  KWY_TRY(kwy_get_poly(ctx, 12ull * (N / syn_pulse_nt<LOG2N>::value), &poly));
  KWY_TRY(get_dc_remover(ctx, N, &dcrem));
  constexpr int NT = syn_pulse_nt<LOG2N>::value;
  size_t lds = sizeof(kwy_c) * ((H + 1) + (H / 8 > 1 ? H / 8 : 1)) +
               sizeof(double) * ((K + 1) + 8) + sizeof(uint32_t) * KWY_EBASE_WORDS;
  KWY_HIP(hipFuncSetAttribute((const void *)k_syn_pulse<LOG2N, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  KWY_HIP(hipFuncSetAttribute((const void *)k_syn_pulse<LOG2N, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // every utterance its slice of a persistent grid (a pulse's slot does not depend on the slice: the waveform
  // is the same whatever the batch)
  const int share = batch.n > 1 ? std::max(64, 4096 / batch.n) : 2048;
  int g = 0;
  for (int u = 0; u < batch.n; ++u) { batch.start[u] = g; g += std::max(1, std::min(batch.u[u].slots, share)); }
  batch.start[batch.n] = g;
  KWY_PROF(ctx, "k_syn_pulse", hipLaunchKernelGGL((k_syn_pulse<LOG2N, false>), dim3(g), dim3(NT), lds, ctx->stream, batch,
                     kwy_randn(ctx), poly, twH, twN, dcrem, (long long *)ctx->dbg));
  g = 0;
  for (int u = 0; u < batch.n; ++u) { batch.start[u] = g; g += batch.u[u].nt; }
  batch.start[batch.n] = g;
  KWY_PROF(ctx, "k_syn_ola", hipLaunchKernelGGL(k_syn_ola, dim3(g), dim3(KWY_THREADS), 0, ctx->stream, batch, N));
  // pulses beyond the slots (none for speech): one workgroup per utterance, serial, same order
  for (int u = 0; u <= batch.n; ++u) batch.start[u] = u;
  KWY_PROF(ctx, "k_syn_pulse_more", hipLaunchKernelGGL((k_syn_pulse<LOG2N, true>), dim3(batch.n), dim3(NT), lds, ctx->stream,
                     batch, kwy_randn(ctx), poly, twH, twN, dcrem, (long long *)nullptr));
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int syn_pulse_cap(int64_t y_length) { return (int)(y_length / 8 + 16); }

static size_t syn_plan_bytes(int64_t y_length);
static size_t syn_plan_scratch_bytes(int64_t y_length);
static size_t syn_render_scratch_bytes(int64_t y_length, int fft_size, int fs) {
  return kwy_pad(sizeof(double) * (size_t)SYN_SLOTS(y_length, fs) * fft_size) + kwy_pad(64);
}
static size_t syn_scratch_bytes(int64_t y_length, int fft_size, int fs) {
  return syn_plan_bytes(y_length) + syn_plan_scratch_bytes(y_length) + syn_render_scratch_bytes(y_length, fft_size, fs);
}

static size_t syn_plan_bytes(int64_t y_length) {
  const int64_t nt = (y_length + SYN_TILE - 1) / SYN_TILE;
  const int cap = syn_pulse_cap(y_length);
  return kwy_pad(64) + kwy_pad(sizeof(int) * (nt + 1)) + kwy_pad(y_length) + kwy_pad(sizeof(int32_t) * cap) +
         kwy_pad(sizeof(double) * cap);
}

static syn_plan syn_plan_carve(void *buffer, int64_t y_length) {
  const int64_t nt = (y_length + SYN_TILE - 1) / SYN_TILE;
  const int cap = syn_pulse_cap(y_length);
  char *q = (char *)buffer;
  syn_plan pl;
  pl.npulse = (int *)q; q += kwy_pad(64);
  pl.tile_cnt = (int *)q; q += kwy_pad(sizeof(int) * (nt + 1));
  pl.vuv8 = (unsigned char *)q; q += kwy_pad(y_length);
  pl.pidx = (int32_t *)q; q += kwy_pad(sizeof(int32_t) * cap);
  pl.pshift = (double *)q;
  return pl;
}

static int syn_make_params(kwy_ctx *ctx, int64_t T, int fft_size, double frame_period_ms, int fs, double sp_mul,
                           int64_t y_length, syn_params *p, int *log2n) {
  *log2n = kwy_ilog2(fft_size);
  if ((1 << *log2n) != fft_size || *log2n < 9 || *log2n > 13) {
    ctx->err = "synthesize: fft_size must be a power of two in [512, 8192]";
    return KWY_EINVAL;
  }
  p->T = T; p->y_length = y_length; p->fs = fs; p->fft_size = fft_size;
  p->frame_period = frame_period_ms / 1000.0;
  p->lowest_f0 = fs / fft_size + 1.0;  // integer division, as upstream
  p->sp_mul = sp_mul;
  return KWY_OK;
}

// scratch of the placement alone (phase scan): from the context's arena
static size_t syn_plan_scratch_bytes(int64_t y_length) {
  return 2 * kwy_pad(sizeof(double) * y_length) + 5 * kwy_pad(8 * (y_length / 4096 + 2)) + kwy_pad(64) +
         kwy_pad(sizeof(syn_segs) * (y_length / 4096 + 2)) + kwy_pad(sizeof(syn_chunks) * (y_length / 4096 + 2));
}

// the scratch of one utterance's placement from the context's arena into `v`
static int syn_plan_fill(kwy_ctx *ctx, const double *f0, const syn_params &p, const syn_plan &pl, syn_plan_view *v) {
  const int64_t y_length = p.y_length;
  const size_t npt_alloc = (size_t)((y_length + SYN_PH_TILE - 1) / SYN_PH_TILE) + 1;
  v->f0 = f0; v->p = p; v->pl = pl;
  v->cap = syn_pulse_cap(y_length);
  v->incr = kwy_arena<double>(ctx, y_length);
  v->tsum = kwy_arena<double>(ctx, npt_alloc);
  v->tin = kwy_arena<double>(ctx, npt_alloc);
  v->summ = kwy_arena<long long>(ctx, 3 * npt_alloc);
  v->segs = kwy_arena<syn_segs>(ctx, npt_alloc);
  v->chunks = kwy_arena<syn_chunks>(ctx, npt_alloc);
  v->wrap = kwy_arena<double>(ctx, y_length);
  if (!v->incr || !v->tsum || !v->tin || !v->summ || !v->segs || !v->chunks || !v->wrap) {
    ctx->err = "synthesize: scratch arena too small";
    return KWY_ENOMEM;
  }
  return KWY_OK;
}

// the placement of the batch's utterances: eight launches whatever their number
static int synth_plan_launch(kwy_ctx *ctx, syn_plan_batch &b) {
  int64_t ymax = 0;
  for (int u = 0; u < b.n; ++u) ymax = std::max(ymax, b.u[u].p.y_length);
  for (int u = b.n; u < KWY_BATCH_MAX; ++u) b.u[u] = b.u[0];
  b.dbg = (long long *)ctx->dbg;
  const unsigned n = (unsigned)b.n;
  const unsigned npt = (unsigned)((ymax + SYN_PH_TILE - 1) / SYN_PH_TILE), nt = (unsigned)((ymax + SYN_TILE - 1) / SYN_TILE);
  hipLaunchKernelGGL(k_syn_inc, dim3((unsigned)((ymax + KWY_THREADS - 1) / KWY_THREADS), n), dim3(KWY_THREADS), 0,
                     ctx->stream, b);
  hipLaunchKernelGGL(k_syn_tile_sums, dim3(npt, n), dim3(KWY_THREADS), 0, ctx->stream, b);
  hipLaunchKernelGGL(k_syn_tile_summary, dim3(npt, n), dim3(KWY_THREADS), 0, ctx->stream, b);
  KWY_PROF(ctx, "k_syn_phase", hipLaunchKernelGGL(k_syn_phase, dim3(n), dim3(SYN_PH_THREADS), 0, ctx->stream, b));
  hipLaunchKernelGGL(k_syn_tile_apply, dim3(npt, n), dim3(KWY_THREADS), 0, ctx->stream, b);
  hipLaunchKernelGGL(k_syn_pulse_count, dim3(nt, n), dim3(KWY_THREADS), 0, ctx->stream, b);
  hipLaunchKernelGGL(k_syn_scan_counts, dim3(n), dim3(KWY_THREADS), 0, ctx->stream, b);
  hipLaunchKernelGGL(k_syn_pulse_emit, dim3(nt, n), dim3(KWY_THREADS), 0, ctx->stream, b);
  KWY_HIP(hipGetLastError());
  return KWY_OK;
}

static int synth_plan(kwy_ctx *ctx, const double *f0, const syn_params &p, const syn_plan &pl) {
  syn_plan_batch b;
  b.n = 1;
  KWY_TRY(syn_plan_fill(ctx, f0, p, pl, &b.u[0]));
  return synth_plan_launch(ctx, b);
}

static int syn_fill_view(kwy_ctx *ctx, const syn_plan &pl, const double *sp, const double *ap, const syn_params &p,
                         double *y, syn_view *v) {
  const int64_t y_length = p.y_length;
  v->sp = sp; v->ap = ap; v->p = p;
  v->pidx = pl.pidx; v->pshift = pl.pshift; v->vuv8 = pl.vuv8; v->npulse = pl.npulse; v->tile_off = pl.tile_cnt;
  v->cap = syn_pulse_cap(y_length);
  v->slots = SYN_SLOTS(y_length, p.fs);
  v->nt = (int)((y_length + SYN_TILE - 1) / SYN_TILE);
  v->resp = kwy_arena<double>(ctx, (size_t)v->slots * p.fft_size);
  v->y = y;
  if (!v->resp) { ctx->err = "synthesize: scratch arena too small"; return KWY_ENOMEM; }
  return KWY_OK;
}

static int syn_launch(kwy_ctx *ctx, syn_batch &batch, int log2n) {
  switch (log2n) {
    case 9: return launch_pulse<9>(ctx, batch);
    case 10: return launch_pulse<10>(ctx, batch);
    case 11: return launch_pulse<11>(ctx, batch);
    case 12: return launch_pulse<12>(ctx, batch);
    // 8192: features resampled up to 96 kHz (3078 bins -> 4097, kwiiyatta/vocoder/world.py:71-78); rare, runs
    // with the 256-thread layout of the shorter transforms (register spills accepted)
    default: return launch_pulse<13>(ctx, batch);
  }
}

static int synth_render(kwy_ctx *ctx, const syn_plan &pl, const double *sp, const double *ap, const syn_params &p,
                        int log2n, double *y) {
  syn_batch batch;
  batch.n = 1;
  KWY_TRY(syn_fill_view(ctx, pl, sp, ap, p, y, &batch.u[0]));
  return syn_launch(ctx, batch, log2n);
}

static int synth_core(kwy_ctx *ctx, const double *f0, int64_t T, const double *sp, const double *ap,
                      int fft_size, double frame_period_ms, int fs, double sp_mul, int64_t y_length,
                      double *y) {
  syn_params p;
  int log2n;
  KWY_TRY(syn_make_params(ctx, T, fft_size, frame_period_ms, fs, sp_mul, y_length, &p, &log2n));
  if (y_length < 2 || T < 2) {       // nothing to place: silence (otherwise the overlap-add writes every sample)
    KWY_HIP(hipMemsetAsync(y, 0, sizeof(double) * y_length, ctx->stream));
    return KWY_OK;
  }
  void *buffer = kwy_arena_alloc(ctx, syn_plan_bytes(y_length));
  if (!buffer) { ctx->err = "synthesize: scratch arena too small"; return KWY_ENOMEM; }
  const syn_plan pl = syn_plan_carve(buffer, y_length);
  KWY_TRY(synth_plan(ctx, f0, p, pl));
  return synth_render(ctx, pl, sp, ap, p, log2n, y);
}

static int syn_check(kwy_ctx *ctx, const void *f0, int64_t T, const void *sp, const void *ap,
                     int fft_size, double frame_period_ms, int fs, int64_t y_length, const void *y) {
  if (!ctx) return KWY_EINVAL;
  if (!f0 || !sp || !ap || !y || T <= 0 || fs <= 0 || fft_size <= 0 || !(frame_period_ms > 0) ||
      y_length < 0 || y_length > 0x7fffffff) {
    ctx->err = "synthesize: bad argument";
    return KWY_EINVAL;
  }
  return KWY_OK;
}

extern "C" int kwy_synthesize_dev(kwy_ctx *ctx, const double *f0, int64_t T, const double *sp,
                                  const double *ap, int fft_size, double frame_period_ms, int fs,
                                  double sp_mul, int64_t y_length, double *y) {
  KWY_TRY(syn_check(ctx, f0, T, sp, ap, fft_size, frame_period_ms, fs, y_length, y));
  KWY_HIP(hipSetDevice(ctx->device));
  if (y_length == 0) return KWY_OK;
  KWY_TRY(kwy_arena_begin(ctx, syn_scratch_bytes(y_length, fft_size, fs)));
  return synth_core(ctx, f0, T, sp, ap, fft_size, frame_period_ms, fs, sp_mul, y_length, y);
}

extern "C" int64_t kwy_synth_plan_bytes(int64_t y_length) {
  return y_length >= 0 && y_length <= 0x7fffffff ? (int64_t)syn_plan_bytes(y_length) : 0;
}

extern "C" int kwy_synth_plan_dev(kwy_ctx *ctx, const double *f0, int64_t T, int fft_size, double frame_period_ms,
                                  int fs, int64_t y_length, void *plan) {
  KWY_TRY(syn_check(ctx, f0, T, f0, f0, fft_size, frame_period_ms, fs, y_length, f0));
  if (!plan) { ctx->err = "synth_plan: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  syn_params p;
  int log2n;
  KWY_TRY(syn_make_params(ctx, T, fft_size, frame_period_ms, fs, 1.0, y_length, &p, &log2n));
  if (y_length < 2 || T < 2) return KWY_OK;
  KWY_TRY(kwy_arena_begin(ctx, syn_plan_scratch_bytes(y_length)));
  return synth_plan(ctx, f0, p, syn_plan_carve(plan, y_length));
}

extern "C" int kwy_synth_plan_batch_dev(kwy_ctx *ctx, const kwy_synth_plan_job *jobs, int count, int fft_size,
                                        double frame_period_ms, int fs) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0) { ctx->err = "synth_plan_batch: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  size_t bytes = 0;
  for (int j = 0; j < count; ++j) {
    const kwy_synth_plan_job &q = jobs[j];
    KWY_TRY(syn_check(ctx, q.f0, q.f0_length, q.f0, q.f0, fft_size, frame_period_ms, fs, q.y_length, q.f0));
    if (!q.plan) { ctx->err = "synth_plan_batch: bad argument"; return KWY_EINVAL; }
    bytes += syn_plan_scratch_bytes(q.y_length);
  }
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  syn_plan_batch b;
  b.n = 0;
  for (int j = 0; j < count; ++j) {
    const kwy_synth_plan_job &q = jobs[j];
    syn_params p;
    int log2n;
    KWY_TRY(syn_make_params(ctx, q.f0_length, fft_size, frame_period_ms, fs, 1.0, q.y_length, &p, &log2n));
    if (q.y_length < 2 || q.f0_length < 2) continue;
    KWY_TRY(syn_plan_fill(ctx, q.f0, p, syn_plan_carve(q.plan, q.y_length), &b.u[b.n]));
    if (++b.n == KWY_BATCH_MAX) { KWY_TRY(synth_plan_launch(ctx, b)); b.n = 0; }
  }
  if (b.n > 0) KWY_TRY(synth_plan_launch(ctx, b));
  return KWY_OK;
}

extern "C" int kwy_synth_render_dev(kwy_ctx *ctx, const void *plan, int64_t T, const double *sp, const double *ap,
                                    int fft_size, double frame_period_ms, int fs, double sp_mul, int64_t y_length,
                                    double *y) {
  KWY_TRY(syn_check(ctx, sp, T, sp, ap, fft_size, frame_period_ms, fs, y_length, y));
  if (!plan) { ctx->err = "synth_render: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  if (y_length == 0) return KWY_OK;
  syn_params p;
  int log2n;
  KWY_TRY(syn_make_params(ctx, T, fft_size, frame_period_ms, fs, sp_mul, y_length, &p, &log2n));
  if (y_length < 2 || T < 2) {
    KWY_HIP(hipMemsetAsync(y, 0, sizeof(double) * y_length, ctx->stream));
    return KWY_OK;
  }
  KWY_TRY(kwy_arena_begin(ctx, syn_render_scratch_bytes(y_length, fft_size, fs)));
  return synth_render(ctx, syn_plan_carve(const_cast<void *>(plan), y_length), sp, ap, p, log2n, y);
}

// The rendering of several utterances (include/kwy.h).  A pass of launches takes SYN_RENDER_GROUP utterances -- four
// 10 s utterances are ~9 000 pulses: a dozen rounds of the chip's resident workgroups -- and the passes of a call share
// ONE pool of response slots (the launches of a stream run one after the other: a pass's responses are dead when the
// next pass writes its own).  Until round 4 a call held the slots of all its utterances at once: 116 MB per 10 s
// utterance, 3.7 GB for the 32 pairs of the benchmark; now 0.46 GB per context whatever the number of utterances.
#define SYN_RENDER_GROUP 4
extern "C" int kwy_synth_render_batch_dev(kwy_ctx *ctx, const kwy_synth_job *jobs, int count, int fft_size,
                                          double frame_period_ms, int fs, double sp_mul) {
  if (!ctx) return KWY_EINVAL;
  if (!jobs || count < 0) { ctx->err = "synth_render_batch: bad argument"; return KWY_EINVAL; }
  KWY_HIP(hipSetDevice(ctx->device));
  size_t bytes = 0, group = 0;
  int in_group = 0;
  for (int j = 0; j < count; ++j) {
    const kwy_synth_job &q = jobs[j];
    KWY_TRY(syn_check(ctx, q.spectrogram, q.f0_length, q.spectrogram, q.aperiodicity, fft_size, frame_period_ms, fs,
                      q.y_length, q.y));
    if (!q.plan) { ctx->err = "synth_render_batch: bad argument"; return KWY_EINVAL; }
    if (q.y_length < 2 || q.f0_length < 2) continue;
    group += syn_render_scratch_bytes(q.y_length, fft_size, fs);
    if (++in_group == SYN_RENDER_GROUP) { bytes = std::max(bytes, group); group = 0; in_group = 0; }
  }
  bytes = std::max(bytes, group);
  KWY_TRY(kwy_arena_begin(ctx, bytes));
  syn_batch batch;
  batch.n = 0;
  int log2n = 0;
  for (int j = 0; j < count; ++j) {
    const kwy_synth_job &q = jobs[j];
    if (q.y_length == 0) continue;
    syn_params p;
    KWY_TRY(syn_make_params(ctx, q.f0_length, fft_size, frame_period_ms, fs, sp_mul, q.y_length, &p, &log2n));
    if (q.y_length < 2 || q.f0_length < 2) {
      KWY_HIP(hipMemsetAsync(q.y, 0, sizeof(double) * q.y_length, ctx->stream));
      continue;
    }
    KWY_TRY(syn_fill_view(ctx, syn_plan_carve(const_cast<void *>(q.plan), q.y_length), q.spectrogram, q.aperiodicity, p,
                          q.y, &batch.u[batch.n]));
    if (++batch.n == SYN_RENDER_GROUP) {
      KWY_TRY(syn_launch(ctx, batch, log2n));
      batch.n = 0;
      ctx->arena_off = 0;               // the next pass takes the same slots
    }
  }
  if (batch.n > 0) KWY_TRY(syn_launch(ctx, batch, log2n));
  return KWY_OK;
}

extern "C" int kwy_synthesize(kwy_ctx *ctx, const double *f0, int64_t T, const double *sp,
                              const double *ap, int fft_size, double frame_period_ms, int fs,
                              double sp_mul, int64_t y_length, double *y) {
  KWY_TRY(syn_check(ctx, f0, T, sp, ap, fft_size, frame_period_ms, fs, y_length, y));
  KWY_HIP(hipSetDevice(ctx->device));
  if (y_length == 0) return KWY_OK;
  for (int64_t i = 0; i < T; ++i)
    if (!(f0[i] >= 0.0 && f0[i] < fs / 12.0)) { ctx->err = "synthesize: f0 must lie in [0, fs/12)"; return KWY_EINVAL; }
  const int K = fft_size / 2 + 1;
  size_t bf = kwy_pad(sizeof(double) * T), bs = kwy_pad(sizeof(double) * T * K);
  size_t by = kwy_pad(sizeof(double) * y_length);
  KWY_TRY(kwy_arena_begin(ctx, syn_scratch_bytes(y_length, fft_size, fs) + bf + 2 * bs + by));
  double *df0 = kwy_arena<double>(ctx, T), *dsp = kwy_arena<double>(ctx, (size_t)T * K);
  double *dap = kwy_arena<double>(ctx, (size_t)T * K), *dy = kwy_arena<double>(ctx, y_length);
  KWY_HIP(hipMemcpyAsync(df0, f0, sizeof(double) * T, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dsp, sp, sizeof(double) * T * K, hipMemcpyHostToDevice, ctx->stream));
  KWY_HIP(hipMemcpyAsync(dap, ap, sizeof(double) * T * K, hipMemcpyHostToDevice, ctx->stream));
  KWY_TRY(synth_core(ctx, df0, T, dsp, dap, fft_size, frame_period_ms, fs, sp_mul, y_length, dy));
  KWY_HIP(hipMemcpyAsync(y, dy, sizeof(double) * y_length, hipMemcpyDeviceToHost, ctx->stream));
  KWY_HIP(hipStreamSynchronize(ctx->stream));
  return KWY_OK;
}
