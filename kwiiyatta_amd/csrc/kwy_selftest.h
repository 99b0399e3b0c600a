/* kwy_selftest.h -- NOT part of the public ABI (include/kwy.h): entry points that exist only so that tests can
 * drive device-side building blocks on caller data. */
#ifndef KWY_SELFTEST_H_
#define KWY_SELFTEST_H_
#include "../../include/kwy.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The block-wide "sum of the m smallest of n" used by D4C's band aperiodicity.
 * values: problems x n non-negative doubles (device); out: problems x {sum of the m smallest, sum of all} (device).
 * n <= 2304. */
int kwy_debug_smallest_sum_dev(kwy_ctx *ctx, const double *values, int problems, int n, int m, double *out);
#ifdef __cplusplus
}
#endif
#endif
