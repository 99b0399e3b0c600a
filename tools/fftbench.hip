// micro-benchmark: LDS FFT variants, cycles per 2048-point complex FFT per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../kwiiyatta_amd/csrc/kwy_device.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Round 4 experiment, measured and NOT adopted: a wave-local 2048-point transform with two
// workgroup barriers instead of seven.  Alone it is 10-12 % faster than the in-place radix-8 transform at the
// occupancies of the D4C kernels (1436 vs 1624 ns per transform and CU at three workgroups per CU, 1370 vs 1521 at
// four); inside k_d4c_body / k_d4c_bands / k_d4c_lovetrain it changed nothing (bands -1 %, LoveTrain -2 %) or cost
// (body +4.5 %: 84 bytes per lane of spills at its register cap) -- the barrier stalls of one workgroup's transform
// are already covered by the other resident workgroups' non-FFT phases.
// ------------------------------------------------- wave-local FFT (round 4): 2048 points, 256 threads
// The in-place transform above meets at a workgroup barrier twice per radix-8 pass: seven barriers for 2048 points, and
// with three or four workgroups per CU the wavefronts spend a third of their cycles waiting.  Here the four wavefronts
// of the workgroup each transform ONE 512-point subsequence on their own -- x[w + 4 m], decimation in time across the
// wavefronts -- three radix-8 stages whose operands change lanes through the wavefront's PRIVATE quarter of the
// buffer (the LDS executes one wavefront's instructions in order: no barrier), and only the closing radix-4 stage
// across the wavefronts needs the workgroup: TWO barriers per transform.
//
//   stage 1   lane l, slot t holds f[l + 64 t]: 8-point DFT over the slots, times W512^(l t')
//   stage 2   (lane a + 8 b, slot t') -> (lane a + 8 t', slot b); DFT over b, times W64^(a b')
//   stage 3   (lane a + 8 t', slot b') -> (lane b' + 8 t', slot a); DFT over a: F[64 a' + 8 b' + t'],
//             times W2048^(w k') for the radix-4 across the wavefronts
//   tail      X[k' + 512 q] = sum_w (-i)^(w q) G_w[k'], in place
//
// Layouts (complex indices; a wavefront's region is 516 entries: the four spare ones keep the producers' stores and the
// tail off each other's banks):
//   input    packed point n at region n & 3, entry 64 (n >> 8) + ((n >> 2) & 63)      kwy_fftw_in()
//   output   X[k] at region k >> 9, entry (k & 511) ^ ((k >> 3) & 7)                  kwy_fftw_at()
// Every store and load below is free of bank conflicts (b128: 16 lanes per cycle over 16 x 16 bytes).
#define KWY_FFTW_REGION 516
#define KWY_FFTW_ENTRIES (4 * KWY_FFTW_REGION)       // complex entries of the buffer (2048-point transform)
__device__ __forceinline__ int kwy_fftw_in(int n) {              // packed complex point n of the input
  return KWY_FFTW_REGION * (n & 3) + 64 * (n >> 8) + ((n >> 2) & 63);
}
__device__ __forceinline__ int kwy_fftw_in_real(int i) {         // real sample i (packed two per point): index in doubles
  return 2 * kwy_fftw_in(i >> 1) + (i & 1);
}
__device__ __forceinline__ int kwy_fftw_at(int k) {              // bin k of the output
  return KWY_FFTW_REGION * (k >> 9) + ((k & 511) ^ ((k >> 3) & 7));
}
struct kwy_fftw_tw { kwy_c b1, b2, b3, cw; };
// twH: exp(-2 pi i k / 2048), k < 2048
__device__ __forceinline__ kwy_fftw_tw kwy_fftw_twiddles(const kwy_c *__restrict__ twH) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  kwy_fftw_tw t;
  t.b1 = twH[4 * l];                                    // W512^l
  t.b2 = twH[32 * (l & 7)];                             // W64^(l & 7)
  t.b3 = twH[(w * (8 * (l & 7) + (l >> 3))) & 2047];    // W2048^(w (8 b' + t')), lane = b' + 8 t'
  t.cw = twH[64 * __builtin_amdgcn_readfirstlane(w)];   // W2048^(64 w): the same for the whole wavefront (scalar registers)
  return t;
}
__device__ __forceinline__ kwy_c kwy_opaque_c(kwy_c v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y));
  return v;
}
// a[m] *= base^m (m = 1..7), the powers formed by multiplication
__device__ __forceinline__ void kwy_fftw_scale(kwy_c (&a)[8], kwy_c w1) {
  const kwy_c w2 = cmulf(w1, w1), w4 = cmulf(w2, w2);
  const kwy_c w3 = cmulf(w1, w2), w5 = cmulf(w4, w1), w6 = cmulf(w4, w2);
  const kwy_c w7 = cmulf(w4, w3);
  a[1] = cmulf(w1, a[1]); a[2] = cmulf(w2, a[2]); a[3] = cmulf(w3, a[3]); a[4] = cmulf(w4, a[4]);
  a[5] = cmulf(w5, a[5]); a[6] = cmulf(w6, a[6]); a[7] = cmulf(w7, a[7]);
}
// orders this wavefront's LDS stores before its following LDS loads (for the compiler; the hardware keeps the order)
__device__ __forceinline__ void kwy_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Forward transform of 2048 packed points (256 threads).  z: KWY_FFTW_ENTRIES entries, input in the kwy_fftw_in layout
// and complete (a barrier between the producers' stores and this call); output in the kwy_fftw_at layout; ends with a
// barrier.  Unnormalised.
// everything behind stage 1's butterfly: a[t'] = the 8-point DFT over the slots of this lane
__device__ __forceinline__ void kwy_fftw_2048_rest(kwy_c *z, const kwy_fftw_tw &tw, kwy_c (&a)[8]) {
  const int tid = kwy_tid_opaque();
  const int l = tid & 63, w = tid >> 6;
  kwy_c *R = z + KWY_FFTW_REGION * w;
  kwy_fftw_scale(a, kwy_opaque_c(tw.b1));
  {
    const int la = l & 7, lb = l >> 3;           // lane = a + 8 b
#pragma unroll
    for (int t = 0; t < 8; ++t) R[64 * lb + ((la + 8 * t) ^ (8 * (lb & 1)))] = a[t];
  }
  kwy_wave_lds_sync();
  // ---- stage 2: lane = a + 8 t', slots b
#pragma unroll
  for (int s = 0; s < 8; ++s) a[s] = R[64 * s + (l ^ (8 * (s & 1)))];
  kwy_dft8<false>(a);
  kwy_fftw_scale(a, kwy_opaque_c(tw.b2));
  {
    const int la = l & 7, lt = l >> 3;           // lane = a + 8 t'
#pragma unroll
    for (int b = 0; b < 8; ++b) R[64 * la + ((b + 8 * lt) ^ la)] = a[b];
  }
  kwy_wave_lds_sync();
  // ---- stage 3: lane = b' + 8 t', slots a
#pragma unroll
  for (int s = 0; s < 8; ++s) a[s] = R[64 * s + (l ^ s)];
  kwy_dft8<false>(a);
  {
    // G[k'] = W2048^(w k') F[k'], k' = 64 a' + 8 b' + t': b3 times cw^a'
    kwy_c p = kwy_opaque_c(tw.b3);
    const kwy_c c = kwy_opaque_c(tw.cw);
    const int lb = l & 7, lt = l >> 3;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      R[64 * m + 8 * lb + (lt ^ lb)] = cmulf(p, a[m]);
      p = cmulf(p, c);
    }
  }
  __syncthreads();
  // ---- the radix-4 across the wavefronts, in place: two butterflies per thread
  kwy_c g[2][4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int k = tid + 256 * it;
    const int e = k ^ ((k >> 3) & 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[it][q] = z[KWY_FFTW_REGION * q + e];
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int k = tid + 256 * it;
    const int e = k ^ ((k >> 3) & 7);
    const kwy_c apc = cadd(g[it][0], g[it][2]), amc = csub(g[it][0], g[it][2]);
    const kwy_c bpd = cadd(g[it][1], g[it][3]), jb = kwy_rot90<false>(csub(g[it][1], g[it][3]));
    z[e] = cadd(apc, bpd);
    z[KWY_FFTW_REGION + e] = cadd(amc, jb);
    z[2 * KWY_FFTW_REGION + e] = csub(apc, bpd);
    z[3 * KWY_FFTW_REGION + e] = csub(amc, jb);
  }
  __syncthreads();
}
__device__ inline void kwy_fftw_2048(kwy_c *z, const kwy_fftw_tw &tw) {
  const int tid = kwy_tid_opaque();
  const kwy_c *R = z + KWY_FFTW_REGION * (tid >> 6);
  kwy_c a[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) a[t] = R[64 * t + (tid & 63)];
  kwy_dft8<false>(a);
  kwy_fftw_2048_rest(z, tw, a);
}
// The same transform of an input that is zero from point 257 on, handed over in registers: thread (w, l) supplies
// point n = w + 4 l in f0, thread 0 also point 256 in f1 (zero elsewhere).  Stage 1's butterfly is a copy -- nothing is
// read, and no barrier is needed in front, as long as the buffer is free.
__device__ inline void kwy_fftw_2048_sparse(kwy_c *z, const kwy_fftw_tw &tw, kwy_c f0, kwy_c f1) {
  kwy_c a[8];
  if (kwy_tid_opaque() == 0) {
    a[0] = f0; a[1] = f1;
#pragma unroll
    for (int m = 2; m < 8; ++m) a[m] = {0.0, 0.0};
    kwy_dft8<false>(a);
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m) a[m] = f0;
  }
  kwy_fftw_2048_rest(z, tw, a);
}
// bin k of the REAL transform whose packed half-length transform kwy_fftw_2048 left in z (cf. kwy_rfft_bin2_w:
// twice the bin, w = exp(-2 pi i k / 4096))
__device__ __forceinline__ kwy_c kwy_fftw_rbin2(const kwy_c *z, int k, kwy_c w) {
  constexpr int H = 2048;
  if (k == 0) return {2.0 * (z[0].x + z[0].y), 0.0};
  if (k == H) return {2.0 * (z[0].x - z[0].y), 0.0};
  const kwy_c A = z[kwy_fftw_at(k)];
  const kwy_c Bc = z[kwy_fftw_at(H - k)];
  const kwy_c B = {Bc.x, -Bc.y};
  const double er = A.x + B.x, ei = A.y + B.y;
  const double dr = A.x - B.x, di = A.y - B.y;
  return {__builtin_fma(dr, w.y, __builtin_fma(di, w.x, er)), __builtin_fma(-dr, w.x, __builtin_fma(di, w.y, ei))};
}


// ---------------------------------------------------------------------------------------------------------------
// (Measured, round 4, and NOT adopted either: 883 ns per transform and CU against the in-place transform's 809 at five
// workgroups per CU, 955 against 953 at three -- radix-8 passes with half the lanes idle still beat five radix-4
// stages with all lanes busy: fewer twiddle products and fewer LDS round trips per point.)
// 1024 points on 256 threads.  The in-place radix-8 transform has 128 butterflies per pass there: two of the four
// wavefronts -- two of the CU's four SIMDs -- sit the passes out.  The wave-local form gives every lane one radix-4
// butterfly per stage: four stages inside a wavefront (its 256-point subsequence x[w + 4 m], operands changing lanes
// through the wavefront's own quarter of the buffer, no barrier), then the radix-4 across the wavefronts.
// Drop-in: natural order in and out, `tw` = exp(-2 pi i k / 1024), k < 128 (the table the callers keep in LDS).
//   lane l = a + 4 b + 16 c, slot t: m = l + 64 t
//   stage 1  DFT over t, times W256^(l t')            -> (lane a + 4 b + 16 t', slot c)
//   stage 2  DFT over c, times W64^((a + 4 b) c')     -> (lane a + 4 c' + 16 t', slot b)
//   stage 3  DFT over b, times W16^(a b')             -> (lane b' + 4 c' + 16 t', slot a)
//   stage 4  DFT over a: F[64 a' + 16 b' + 4 c' + t'], times W1024^(w k')
__device__ __forceinline__ kwy_c fftw_tw1024(const kwy_c *__restrict__ tw, int i) {   // exp(-2 pi i i / 1024), i < 1024
  return kwy_tw_octant(tw[i & 127], i >> 7);
}
template <bool INV>
__device__ __forceinline__ void fftw_dft4(kwy_c (&a)[4]) {
  const kwy_c apc = cadd(a[0], a[2]), amc = csub(a[0], a[2]);
  const kwy_c bpd = cadd(a[1], a[3]), jb = kwy_rot90<INV>(csub(a[1], a[3]));
  a[0] = cadd(apc, bpd); a[1] = cadd(amc, jb); a[2] = csub(apc, bpd); a[3] = csub(amc, jb);
}
template <bool INV>
__device__ __forceinline__ void fftw_scale4(kwy_c (&a)[4], kwy_c w1) {
  if (INV) w1.y = -w1.y;
  const kwy_c w2 = cmulf(w1, w1), w3 = cmulf(w1, w2);
  a[1] = cmulf(w1, a[1]); a[2] = cmulf(w2, a[2]); a[3] = cmulf(w3, a[3]);
}
__device__ __forceinline__ void fftw_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool INV>
__device__ inline void kwy_fftw_1024(kwy_c *z, const kwy_c *__restrict__ tw) {
  const int tid = kwy_tid_opaque();
  const int l = tid & 63, w = tid >> 6;
  kwy_c *R = z + 256 * w;
  kwy_c a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[t] = z[w + 4 * l + 256 * t];       // (the one access with a bank conflict: 4-way)
  __syncthreads();                                                 // every wavefront has its inputs: the regions are free
  // ---- stage 1
  fftw_dft4<INV>(a);
  fftw_scale4<INV>(a, fftw_tw1024(tw, 4 * l));
  {
    const int ab = l & 15, c = l >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) R[64 * c + ab + 16 * t] = a[t];
  }
  fftw_wave_sync();
  // ---- stage 2: lane = a + 4 b + 16 t', slots c
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s] = R[64 * s + l];
  fftw_dft4<INV>(a);
  fftw_scale4<INV>(a, fftw_tw1024(tw, 16 * (l & 15)));
  {
    const int la = l & 3, lb = (l >> 2) & 3, lt = l >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) R[64 * lb + ((la + 4 * c + 16 * lt) ^ (4 * lb))] = a[c];
  }
  fftw_wave_sync();
  // ---- stage 3: lane = a + 4 c' + 16 t', slots b
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s] = R[64 * s + (l ^ (4 * s))];
  fftw_dft4<INV>(a);
  fftw_scale4<INV>(a, fftw_tw1024(tw, 64 * (l & 3)));
  {
    const int la = l & 3, lc = (l >> 2) & 3, lt = l >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) R[64 * la + ((b + 4 * lc + 16 * lt) ^ la)] = a[b];
  }
  fftw_wave_sync();
  // ---- stage 4: lane = b' + 4 c' + 16 t', slots a
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s] = R[64 * s + (l ^ s)];
  fftw_dft4<INV>(a);
  {
    // G[k'] = W1024^(w k') F[k'], k' = 64 a' + 16 b' + 4 c' + t'
    const int lb = l & 3, lc = (l >> 2) & 3, lt = l >> 4;
    kwy_c p = fftw_tw1024(tw, w * (16 * lb + 4 * lc + lt));
    kwy_c c = fftw_tw1024(tw, 64 * w);
    if (INV) { p.y = -p.y; c.y = -c.y; }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      R[64 * m + 16 * lb + 4 * lc + (lt ^ lb)] = cmulf(p, a[m]);      // entry k' ^ ((k' >> 4) & 3)
      p = cmulf(p, c);
    }
  }
  __syncthreads();
  // ---- the radix-4 across the wavefronts: X[k' + 256 q], natural order
  {
    const int e = tid ^ ((tid >> 4) & 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = z[256 * q + e];
  }
  fftw_dft4<INV>(a);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) z[256 * q + tid] = a[q];
  __syncthreads();
}

// Baseline for the comparison: the two-buffer radix-4 Stockham transform the
// kernels used before kwy_fft_inplace replaced it.
// One Stockham radix-4 pass over H points: x -> y, sub-transform stride s.
// tw: H-entry table exp(-2 pi i k / H).  INV conjugates the twiddles.
template <bool INV, int NT = KWY_THREADS>
__device__ __forceinline__ void kwy_fft_r4(const kwy_c *__restrict__ x, kwy_c *__restrict__ y,
                                           int H, int s, int log2s,
                                           const kwy_c *__restrict__ tw) {
  const int Q = H >> 2;
  for (int j = threadIdx.x; j < Q; j += NT) {
    const int q = j & (s - 1);
    const int p = j >> log2s;
    kwy_c a = x[j], b = x[j + Q], c = x[j + 2 * Q], d = x[j + 3 * Q];
    kwy_c apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    kwy_c jb = INV ? kwy_c{bmd.y, -bmd.x} : kwy_c{-bmd.y, bmd.x};  // = +-i*(b-d), sign folded below
    // forward: y1 = amc - i*bmd, y3 = amc + i*bmd ; inverse: swapped
    kwy_c y0 = cadd(apc, bpd);
    kwy_c y1 = csub(amc, jb);
    kwy_c y2 = csub(apc, bpd);
    kwy_c y3 = cadd(amc, jb);
    const int ps = p << log2s;
    kwy_c w1 = tw[ps], w2 = tw[2 * ps], w3 = tw[3 * ps];
    if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
    const int o = q + ((4 * p) << log2s);
    y[o] = y0;
    y[o + s] = cmul(w1, y1);
    y[o + 2 * s] = cmul(w2, y2);
    y[o + 3 * s] = cmul(w3, y3);
  }
}

// final radix-2 pass (sub-transform size 2, no twiddle), s = H/2
template <int NT = KWY_THREADS>
__device__ __forceinline__ void kwy_fft_r2(const kwy_c *__restrict__ x, kwy_c *__restrict__ y, int H) {
  const int s = H >> 1;
  for (int q = threadIdx.x; q < s; q += NT) {
    kwy_c a = x[q], b = x[q + s];
    y[q] = cadd(a, b);
    y[q + s] = csub(a, b);
  }
}

// Complex FFT of H = 2^log2H points held in LDS buffer a; b is a scratch buffer
// of the same size.  Returns the buffer that holds the (natural order) result.
// Unnormalised in both directions.  Ends with a barrier.
template <bool INV, int NT = KWY_THREADS>
__device__ inline kwy_c *kwy_fft_lds(kwy_c *a, kwy_c *b, int log2H, const kwy_c *__restrict__ tw) {
  const int H = 1 << log2H;
  kwy_c *src = a, *dst = b;
  int log2s = 0;
  __syncthreads();
  for (int rem = log2H; rem >= 2; rem -= 2) {
    kwy_fft_r4<INV, NT>(src, dst, H, 1 << log2s, log2s, tw);
    __syncthreads();
    kwy_c *t = src; src = dst; dst = t;
    log2s += 2;
  }
  if (log2H & 1) {
    kwy_fft_r2<NT>(src, dst, H);
    __syncthreads();
    kwy_c *t = src; src = dst; dst = t;
  }
  return src;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LOG2H, int NT, int VAR>
__global__ __launch_bounds__(NT) void k_fft(const kwy_c *__restrict__ tw, double *out, int reps, long long *cyc) {
  constexpr int H = 1 << LOG2H;
  extern __shared__ double smem[];
  kwy_c *A = (kwy_c *)smem;
  kwy_c *B2 = A + (H + 1);
  for (int i = threadIdx.x; i < H; i += NT) A[i] = {(double)((i * 37 + blockIdx.x) % 101) - 50.0, (double)((i * 11) % 17) - 8.0};
  __syncthreads();
  long long t0 = clock64();
  kwy_c *r = A;
  for (int it = 0; it < reps; ++it) {
    if (VAR == 0) { kwy_fft_inplace<LOG2H, NT, false>(A, tw); r = A; }
    if (VAR == 1) { r = kwy_fft_lds<false, NT>(A, B2, LOG2H, tw); }
    if constexpr (VAR == 2) { kwy_fftw_2048(A, kwy_fftw_twiddles(tw)); r = A; }
    if constexpr (VAR == 3) { kwy_fftw_1024<false>(A, tw); r = A; }
    if constexpr (VAR == 4) { kwy_fftw_1024<true>(A, tw); r = A; }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  double acc = 0;
  for (int i = threadIdx.x; i < H; i += NT) acc += r[i].x + r[i].y;
  if (acc == 12345.678) out[blockIdx.x] = acc;
}

// the wave-local transform against the in-place one on the same input: max |difference| over the bins
__global__ __launch_bounds__(256) void k_check(const kwy_c *__restrict__ tw, double *out) {
  extern __shared__ double smem[];
  kwy_c *A = (kwy_c *)smem;
  kwy_c *B2 = A + 2049;
  for (int i = threadIdx.x; i < 2048; i += 256) {
    const kwy_c v = {sin(0.37 * i + blockIdx.x) * 3.0 + (i % 7), cos(1.3 * i) - 0.01 * i};
    A[i] = v;
    B2[kwy_fftw_in(i)] = v;
  }
  __syncthreads();
  kwy_fft_inplace<11, 256, false>(A, tw);
  kwy_fftw_2048(B2, kwy_fftw_twiddles(tw));
  double e = 0.0, m = 0.0;
  for (int k = threadIdx.x; k < 2048; k += 256) {
    const kwy_c a = A[k], b = B2[kwy_fftw_at(k)];
    e = fmax(e, fmax(fabs(a.x - b.x), fabs(a.y - b.y)));
    m = fmax(m, fmax(fabs(a.x), fabs(a.y)));
  }
  __shared__ double se[256], sm[256];
  se[threadIdx.x] = e; sm[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) { e = fmax(e, se[i]); m = fmax(m, sm[i]); }
    out[2 * blockIdx.x] = e; out[2 * blockIdx.x + 1] = m;
  }
}

template <bool INV>
__global__ __launch_bounds__(256) void k_check1024(const kwy_c *__restrict__ tw, double *out) {
  extern __shared__ double smem[];
  kwy_c *A = (kwy_c *)smem;
  kwy_c *B2 = A + 1025;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const kwy_c v = {sin(0.37 * i + blockIdx.x) * 3.0 + (i % 7), cos(1.3 * i) - 0.01 * i};
    A[i] = v;
    B2[i] = v;
  }
  __syncthreads();
  kwy_fft_inplace<10, 256, INV>(A, tw);
  kwy_fftw_1024<INV>(B2, tw);
  double e = 0.0, m = 0.0;
  for (int k = threadIdx.x; k < 1024; k += 256) {
    const kwy_c a = A[k], b = B2[k];
    e = fmax(e, fmax(fabs(a.x - b.x), fabs(a.y - b.y)));
    m = fmax(m, fmax(fabs(a.x), fabs(a.y)));
  }
  __shared__ double se[256], sm[256];
  se[threadIdx.x] = e; sm[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) { e = fmax(e, se[i]); m = fmax(m, sm[i]); }
    out[2 * blockIdx.x] = e; out[2 * blockIdx.x + 1] = m;
  }
}

template <int LOG2H, int NT, int VAR>
void run(const char *name, const kwy_c *tw, size_t lds, int grid, int reps) {
  double *out; long long *cyc;
  CK(hipMalloc(&out, 8 * grid)); CK(hipMalloc(&cyc, 64));
  CK(hipFuncSetAttribute((const void *)k_fft<LOG2H, NT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_fft<LOG2H, NT, VAR>), dim3(grid), dim3(NT), lds, 0, tw, out, reps, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_fft<LOG2H, NT, VAR>), dim3(grid), dim3(NT), lds, 0, tw, out, reps, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  double ffts = (double)grid * reps;
  printf("%-44s lds %6zu  %8.3f ms  %7.1f ns/FFT/CU-equivalent(256CU)  wg0 cycles/FFT %lld\n", name, lds, ms,
         ms * 1e6 / (ffts / 256.0), c / reps);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  const int L = 11, H = 1 << L;
  std::vector<kwy_c> h(H);
  for (int k = 0; k < H; ++k) { double a = -2.0 * M_PI * k / H; h[k].x = cos(a); h[k].y = sin(a); }
  kwy_c *tw; CK(hipMalloc(&tw, sizeof(kwy_c) * H)); CK(hipMemcpy(tw, h.data(), sizeof(kwy_c) * H, hipMemcpyHostToDevice));
  {
    double *out; CK(hipMalloc(&out, 16 * 4));
    CK(hipFuncSetAttribute((const void *)k_check, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(k_check, dim3(4), dim3(256), 80 * 1024, 0, tw, out);
    double h8[8]; CK(hipMemcpy(h8, out, 64, hipMemcpyDeviceToHost));
    for (int b = 0; b < 4; ++b) printf("check block %d: max |wave-local - in-place| = %.3e (max |X| = %.3e)\n", b, h8[2 * b], h8[2 * b + 1]);
  }
  kwy_c *tw10;
  {
    std::vector<kwy_c> h10(1024);
    for (int k = 0; k < 1024; ++k) { double a = -2.0 * M_PI * k / 1024; h10[k].x = cos(a); h10[k].y = sin(a); }
    CK(hipMalloc(&tw10, sizeof(kwy_c) * 1024)); CK(hipMemcpy(tw10, h10.data(), sizeof(kwy_c) * 1024, hipMemcpyHostToDevice));
    double *out; CK(hipMalloc(&out, 16 * 4));
    double h8[8];
    CK(hipFuncSetAttribute((const void *)k_check1024<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
    hipLaunchKernelGGL(k_check1024<false>, dim3(4), dim3(256), 40 * 1024, 0, tw10, out);
    CK(hipMemcpy(h8, out, 64, hipMemcpyDeviceToHost));
    for (int b = 0; b < 2; ++b) printf("check 1024 fwd block %d: max |wave-local - in-place| = %.3e (max |X| = %.3e)\n", b, h8[2 * b], h8[2 * b + 1]);
    CK(hipFuncSetAttribute((const void *)k_check1024<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
    hipLaunchKernelGGL(k_check1024<true>, dim3(4), dim3(256), 40 * 1024, 0, tw10, out);
    CK(hipMemcpy(h8, out, 64, hipMemcpyDeviceToHost));
    for (int b = 0; b < 2; ++b) printf("check 1024 inv block %d: max |wave-local - in-place| = %.3e (max |X| = %.3e)\n", b, h8[2 * b], h8[2 * b + 1]);
  }
  const int grid = 2048, reps = 20;
  run<10, 256, 0>("1024: inplace r8 NT256, 5 WG/CU (lds 29K)", tw10, 29 * 1024, 2560, reps);
  run<10, 256, 3>("1024: wave-local NT256, 5 WG/CU (lds 29K)", tw10, 29 * 1024, 2560, reps);
  run<10, 256, 4>("1024: wave-local inverse, 5 WG/CU (lds 29K)", tw10, 29 * 1024, 2560, reps);
  run<10, 256, 0>("1024: inplace r8 NT256, 3 WG/CU (lds 51K)", tw10, 51 * 1024, 2304, reps);
  run<10, 256, 3>("1024: wave-local NT256, 3 WG/CU (lds 51K)", tw10, 51 * 1024, 2304, reps);
  size_t one = sizeof(kwy_c) * (H + 1), two = 2 * one;
  run<11, 512, 0>("inplace r8 NT512, 1 WG/CU (lds 100K)", tw, 100 * 1024, grid, reps);
  run<11, 512, 0>("inplace r8 NT512, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 512, 0>("inplace r8 NT512, 4 WG/CU (lds 33K)", tw, one, grid, reps);
  run<11, 256, 0>("inplace r8 NT256, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 256, 0>("inplace r8 NT256, 4 WG/CU (lds 33K)", tw, one, grid, reps);
  run<11, 256, 0>("inplace r8 NT256, 3 WG/CU (lds 51K)", tw, 51 * 1024, grid, reps);
  run<11, 256, 2>("wave-local NT256, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 256, 2>("wave-local NT256, 3 WG/CU (lds 51K)", tw, 51 * 1024, grid, reps);
  run<11, 256, 2>("wave-local NT256, 4 WG/CU (lds 33K)", tw, sizeof(kwy_c) * KWY_FFTW_ENTRIES + 64, grid, reps);
  run<11, 512, 1>("pingpong r4 NT512, 1 WG/CU (lds 100K)", tw, 100 * 1024, grid, reps);
  run<11, 512, 1>("pingpong r4 NT512, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 256, 1>("pingpong r4 NT256, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  return 0;
}
