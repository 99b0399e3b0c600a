// latency / issue microbenchmark for the FastDTW step's instructions (one wavefront)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double v, double first) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowshr1(double v, double first) {   // row_shr:1 (within 16 lanes)
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x111, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x111, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define N 4096
__global__ void k(double *out, long long *t, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c = b0 * 2, d = b0 * 3;
  long long t0, t1;
  const long long w0 = wall_clock64(), c0 = clock64();
  // 0: dependent add chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = a + b;
  t1 = clock64(); if (threadIdx.x == 0) t[0] = t1 - t0;
  // 1: dependent min chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) { a = fmin(a, b); b = b + 0.0; asm volatile("" : "+v"(a)); }
  t1 = clock64(); if (threadIdx.x == 0) t[1] = t1 - t0;
  // 2: dependent fma chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = __builtin_fma(a, c, b);
  t1 = clock64(); if (threadIdx.x == 0) t[2] = t1 - t0;
  // 3: dependent wave_shr chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = shr1(a, b);
  t1 = clock64(); if (threadIdx.x == 0) t[3] = t1 - t0;
  // 4: 4 independent add chains
  double e = a * 2, f = a * 3, g = a * 4;
  t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { a = a + b; e = e + b; f = f + b; g = g + b; }
  t1 = clock64(); if (threadIdx.x == 0) t[4] = t1 - t0;
  // 5: the step: shr + 3 add + 2 min
  double v1 = a, upp = e, dt = 1e-3;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    const double up = shr1(v1, b);
    const double dg = upp; upp = up;
    v1 = fmin(up + dt, fmin(v1 + dt, dg + dt));
  }
  t1 = clock64(); if (threadIdx.x == 0) t[5] = t1 - t0;
  // 6: row_shr chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = rowshr1(a, b);
  t1 = clock64(); if (threadIdx.x == 0) t[6] = t1 - t0;
  // 7: 4 independent min
  t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { a = fmin(a, d); e = fmin(e, d); f = fmin(f, d); g = fmin(g, d); asm volatile("" : "+v"(a), "+v"(e), "+v"(f), "+v"(g)); }
  t1 = clock64(); if (threadIdx.x == 0) t[7] = t1 - t0;
  // 8: f32 dependent add
  float fa = (float)a, fb = (float)b;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) fa = fa + fb;
  t1 = clock64(); if (threadIdx.x == 0) t[8] = t1 - t0;
  // 9: step with integer-compare min (values non-negative: bit pattern order = value order)
  if (threadIdx.x == 0) { t[10] = wall_clock64() - w0; t[11] = clock64() - c0; }
  out[threadIdx.x] = a + e + f + g + v1 + upp + fa;
}
int main() {
  double *out; long long *t;
  hipMalloc(&out, 64 * 8); hipMalloc(&t, 16 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.0, 1e-9);
  long long h[16]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[] = {"dep add_f64", "dep min_f64(+add)", "dep fma_f64", "dep wave_shr(2 dpp)", "4 indep add_f64 (per iter)", "step shr+3add+2min", "dep row_shr(2 dpp)", "4 indep min", "dep add_f32"};
  for (int i = 0; i < 9; ++i) printf("%-28s %.2f cycles/iter\n", names[i], (double)h[i] / N);
  printf("wall ticks %lld (100 MHz?) clock64 ticks %lld -> clock64 per wall tick %.3f\n", h[10], h[11], (double)h[11]/h[10]);
  return 0;
}
