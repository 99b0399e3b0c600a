// micro-benchmark: LDS FFT variants, cycles per 2048-point complex FFT per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../kwiiyatta_amd/csrc/kwy_device.hpp"

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
