// libkwy_selftest.so -- NOT part of the product: entry points that let tests drive device-side building blocks of
// kwy_device.hpp on caller data.  Built beside libkwy.so by build.sh, loaded only by tests/test_select_gpu.py.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../kwy_device.hpp"
#include "../kwy_selftest.h"

#define D4C_NT 256   // the 256-thread select of the 16..48 kHz band kernel

// ---- diagnostic: the "sum of the m smallest" routine of kwy_device.hpp on caller-supplied data -------------
// (one 256-thread workgroup per problem; out[p] = {sum of the m smallest, sum of all})
__global__ __launch_bounds__(D4C_NT) void k_select_selftest(const double *__restrict__ v, int n, int m,
                                                           double *__restrict__ out) {
  constexpr int RK = 9;
  __shared__ __attribute__((aligned(16))) uint32_t hist[KWY_SELECT_WORDS(D4C_NT)];
  __shared__ double red[2 * D4C_NT / 64];
  const int tid = threadIdx.x;
  const double *x = v + (size_t)blockIdx.x * n;
  unsigned long long key[RK];
#pragma unroll
  for (int r = 0; r < RK; ++r) {
    const int k = tid + D4C_NT * r;
    key[r] = k < n ? (unsigned long long)__double_as_longlong(x[k]) : ~0ull;
  }
  double s_small, s_all;
  kwy_block_smallest_sum<RK, D4C_NT>(key, n, m, hist, red, &s_small, &s_all);
  if (tid == 0) { out[2 * blockIdx.x] = s_small; out[2 * blockIdx.x + 1] = s_all; }
}

extern "C" int kwy_debug_smallest_sum_dev(void *stream, const double *values, int problems, int n, int m,
                                          double *out) {
  if (!values || !out || problems <= 0 || n <= 0 || n > 9 * D4C_NT || m <= 0 || m > n) return -1;
  hipLaunchKernelGGL(k_select_selftest, dim3(problems), dim3(D4C_NT), 0, (hipStream_t)stream, values, n, m, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- diagnostic: kwy_log / kwy_sincos_medium of kwy_device.hpp element by element ----------------------------
__global__ void k_devmath_selftest(const double *__restrict__ x, int n, double *__restrict__ lg, double *__restrict__ sn,
                                   double *__restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lg[i] = kwy_log(x[i]);
  kwy_sincos_medium(x[i], &sn[i], &cs[i]);
}

extern "C" int kwy_debug_devmath_dev(void *stream, const double *x, int n, double *log_out, double *sin_out,
                                     double *cos_out) {
  if (!x || !log_out || !sin_out || !cos_out || n <= 0) return -1;
  hipLaunchKernelGGL(k_devmath_selftest, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n, log_out,
                     sin_out, cos_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
