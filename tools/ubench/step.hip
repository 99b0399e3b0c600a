// the recurrence step of k_dtw_values in isolation: which ingredient costs what (one wavefront per SIMD, 4 waves)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double v, double first) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rol1(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double minraw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define N 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, long long *t, double b0, const double *in) {
  double cur[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) cur[i] = in[i * 64 + (threadIdx.x & 63)];
  double v1 = b0 + threadIdx.x, upp = b0, brot = b0 * 3;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      double up;
      if (MODE & 1) { const double nb = rol1(brot); up = shr1(v1, brot); brot = nb; }
      else up = shr1(v1, brot);
      const double dg = upp; upp = up;
      if (MODE & 2) v1 = minraw(up, minraw(v1, dg)) + cur[u];
      else v1 = fmin(up + cur[u], fmin(v1 + cur[u], dg + cur[u]));
    }
    if (MODE & 4) {   // a store of the 16 values' last one and a bit of scalar bookkeeping
      out[(size_t)i * 256 + threadIdx.x] = v1;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = v1 + upp + brot;
  if ((threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char *name, double *out, long long *t, double *in) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256), 0, 0, out, t, 1e-9, in);
  long long h[4]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-40s %.1f %.1f %.1f %.1f cycles/step\n", name, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N, (double)h[3] / N);
}
int main() {
  double *out, *in; long long *t;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&t, 64); (void)hipMalloc(&in, 16 * 64 * 8); (void)hipMemset(in, 0, 16 * 64 * 8);
  run<0>("shr + 3 add + 2 fmin", out, t, in);
  run<1>("+ rotation (wave_rol)", out, t, in);
  run<2>("shr + 2 v_min + add", out, t, in);
  run<3>("shr + rot + 2 v_min + add", out, t, in);
  run<7>("shr + rot + 2 v_min + add + store/16", out, t, in);
  return 0;
}
