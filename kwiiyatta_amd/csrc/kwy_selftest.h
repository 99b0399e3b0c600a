/* kwy_selftest.h -- NOT part of the public ABI (include/kwy.h) and not in libkwy.so: entry points of the test-only
 * libkwy_selftest.so (selftest/kwy_selftest.hip), which let tests drive device-side building blocks on caller data. */
#ifndef KWY_SELFTEST_H_
#define KWY_SELFTEST_H_
#ifdef __cplusplus
extern "C" {
#endif
/* The block-wide "sum of the m smallest of n" used by D4C's band aperiodicity.
 * values: problems x n non-negative doubles (device); out: problems x {sum of the m smallest, sum of all} (device).
 * n <= 2304.  stream: a hipStream_t (NULL = the default stream).  Returns 0, -1 (arguments) or -2 (launch failed). */
int kwy_debug_smallest_sum_dev(void *stream, const double *values, int problems, int n, int m, double *out);
/* kwy_log(x) and kwy_sincos_medium(x) of kwy_device.hpp for every element of x (n doubles, device). */
int kwy_debug_devmath_dev(void *stream, const double *x, int n, double *log_out, double *sin_out, double *cos_out);
#ifdef __cplusplus
}
#endif
#endif
