// where do the four wavefronts of a 256-thread workgroup with ~250 VGPRs land, and what does sharing a SIMD cost?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double v, double first) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define N 8192
template <int REGS, bool ACC>
__global__ __launch_bounds__(256) void k(double *out, long long *t, double b0, const double *in) {
  double keep[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) keep[i] = in[i * 64 + (threadIdx.x & 63)];
  if (ACC) asm volatile("v_accvgpr_write_b32 a255, 0");
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  double v1 = b0 + threadIdx.x, upp = b0, dt = 1e-3, b = b0;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    const double up = shr1(v1, b);
    const double dg = upp; upp = up;
    v1 = fmin(up + dt, fmin(v1 + dt, dg + dt));
  }
  const long long t1 = clock64();
  double s = v1 + upp;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += keep[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = t1 - t0; t[2 * (threadIdx.x >> 6) + 1] = hw; }
}
template <int REGS, bool ACC> void run(const char *name, double *out, long long *t, double *in) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<REGS, ACC>), dim3(1), dim3(256), 0, 0, out, t, 1e-9, in);
  long long h[8]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s:", name);
  for (int w = 0; w < 4; ++w) printf("  wave %d: %.1f cyc/step simd %lld cu %lld", w, (double)h[2 * w] / N, (h[2 * w + 1] >> 4) & 3, (h[2 * w + 1] >> 8) & 15);
  printf("\n");
}
int main() {
  double *out, *in; long long *t;
  (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&t, 16 * 8); (void)hipMalloc(&in, 128 * 64 * 8); (void)hipMemset(in, 0, 128 * 64 * 8);
  run<4, false>("few regs        ", out, t, in);
  run<110, false>("~250 vgprs      ", out, t, in);
  run<110, true>("~250 vgprs + acc", out, t, in);
  return 0;
}
