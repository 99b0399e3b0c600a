// micro-benchmark: LDS FFT variants, cycles per 2048-point complex FFT per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../kwiiyatta_amd/csrc/kwy_device.hpp"

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
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  double acc = 0;
  for (int i = threadIdx.x; i < H; i += NT) acc += r[i].x + r[i].y;
  if (acc == 12345.678) out[blockIdx.x] = acc;
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
  const int grid = 2048, reps = 20;
  size_t one = sizeof(kwy_c) * (H + 1), two = 2 * one;
  run<11, 512, 0>("inplace r8 NT512, 1 WG/CU (lds 100K)", tw, 100 * 1024, grid, reps);
  run<11, 512, 0>("inplace r8 NT512, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 512, 0>("inplace r8 NT512, 4 WG/CU (lds 33K)", tw, one, grid, reps);
  run<11, 256, 0>("inplace r8 NT256, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 256, 0>("inplace r8 NT256, 4 WG/CU (lds 33K)", tw, one, grid, reps);
  run<11, 512, 1>("pingpong r4 NT512, 1 WG/CU (lds 100K)", tw, 100 * 1024, grid, reps);
  run<11, 512, 1>("pingpong r4 NT512, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  run<11, 256, 1>("pingpong r4 NT256, 2 WG/CU (lds 70K)", tw, 70 * 1024, grid, reps);
  return 0;
}
