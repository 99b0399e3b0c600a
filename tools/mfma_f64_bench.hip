// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (the denominator of the EM kernels' roofline)
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 (without the flag the accumulators are copied
// through AGPRs every iteration and the loop measures those copies)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int reps, long long *cyc) {
  v4f64 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int wgs, int reps) {
  double *out; long long *cyc;
  CK(hipMalloc(&out, 8 * 256 * wgs)); CK(hipMalloc(&cyc, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, out, reps, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, out, reps, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  double mfmas = (double)wgs * 4 * reps * NACC;
  printf("NACC %2d wgs %4d: %8.3f ms  %7.2f TFLOP/s   wave0: %.1f cycles per MFMA\n", NACC, wgs, ms,
         mfmas * 2048.0 / (ms * 1e-3) / 1e12, (double)c / ((double)reps * NACC));
}

int main() {
  run<1>(256, 20000);
  run<4>(256, 5000);
  run<8>(256, 2500);
  run<8>(512, 2500);
  run<8>(1024, 2500);
  return 0;
}
