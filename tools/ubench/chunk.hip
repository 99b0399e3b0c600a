// the recurrence chunk of k_dtw_values piece by piece: what do the memory instructions of a chunk cost a lone wavefront?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2), aligned(16)));
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
#define NCH 256
template <int MODE>
__global__ __launch_bounds__(256) void k(d2 *out, long long *t, double b0, const d2 *in) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double ring[4][16];
  const d2 *src = in + (size_t)wv * NCH * 8 * 64 + lane;
  d2 *dst = out + (size_t)wv * NCH * 8 * 64 + lane;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 8; ++q) { const d2 v = src[(r * 8 + q) * 64]; ring[r][2 * q] = v.x; ring[r][2 * q + 1] = v.y; }
  double v1 = b0 + threadIdx.x, upp = b0, brot = b0 * 3;
  __syncthreads();
  const long long t0 = clock64();
  for (int c = 0; c < NCH; c += 4) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int ch = c + ph;
      if (MODE & 1) {   // loads of the chunk three ahead
        const d2 *p = src + (size_t)min(ch + 3, NCH - 1) * 8 * 64;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const d2 v = p[q * 64]; ring[(ph + 3) & 3][2 * q] = v.x; ring[(ph + 3) & 3][2 * q + 1] = v.y; }
      }
      double hist[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double nb = rol1(brot);
        const double up = shr1(v1, brot); brot = nb;
        const double dg = upp; upp = up;
        v1 = fmin(up, fmin(v1, dg)) + ring[ph][u];
        hist[u] = v1;
      }
      if (MODE & 2) {   // stores of the chunk
        d2 *p = dst + (size_t)ch * 8 * 64;
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q * 64] = d2{hist[2 * q], hist[2 * q + 1]};
      }
      if (MODE & 4) {   // boundary values of one lane to LDS
        if (lane == 63) {
#pragma unroll
          for (int u = 0; u < 16; ++u) lds[wv * 4096 + ((ch * 16 + u) & 4095)] = hist[u];
        }
      }
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = d2{v1 + upp + brot, 0.0};
  if (lane == 0) t[wv] = t1 - t0;
}
template <int MODE, int NT> void run(const char *name, d2 *out, long long *t, d2 *in) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(NT), 4 * 4096 * 8, 0, out, t, 1e-9, in);
  long long h[4]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s %.1f %.1f %.1f %.1f cycles/step\n", name, (double)h[0] / (NCH * 16), (double)h[1] / (NCH * 16), (double)h[2] / (NCH * 16), (double)h[3] / (NCH * 16));
}
int main() {
  d2 *out, *in; long long *t;
  const size_t bytes = (size_t)4 * NCH * 8 * 64 * 16 + 4096;
  (void)hipMalloc(&out, bytes); (void)hipMalloc(&t, 64); (void)hipMalloc(&in, bytes); (void)hipMemset(in, 0, bytes);
  run<0, 256>("steps only (ring in registers)", out, t, in);
  run<1, 256>("+ 8 loads per chunk", out, t, in);
  run<2, 256>("+ 8 stores per chunk", out, t, in);
  run<4, 256>("+ 16 LDS writes of one lane per chunk", out, t, in);
  run<3, 256>("+ loads + stores", out, t, in);
  run<7, 256>("+ loads + stores + LDS", out, t, in);
  run<1, 64>("ONE wave: + 8 loads per chunk", out, t, in);
  run<2, 64>("ONE wave: + 8 stores per chunk", out, t, in);
  run<3, 64>("ONE wave: loads + stores", out, t, in);
  run<3, 128>("TWO waves: loads + stores", out, t, in);
  return 0;
}
