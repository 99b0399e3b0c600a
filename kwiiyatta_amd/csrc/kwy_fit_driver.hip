// kwy_fit_driver.hip -- GaussianMixture(n_components, covariance_type='full', init_params='kmeans').fit(X) as ONE call.
//
// Replaces sklearn.mixture.GaussianMixture.fit as the reference uses it (kwiiyatta/converter/gmm.py:14-26: one call on
// the joint static + delta training matrix) for a binder that has no Python driver to run.  The numerical blocks are
// the kwy_km_*_dev / kwy_gmm_em_*_dev entry points (kwy_kmeans.hip, kwy_gmmfit.hip); this file is the control flow
// scikit-learn wraps around them, for the rows of ONE rank:
//
//   KMeans(n_clusters=M, n_init=1, random_state=seed)     centred data, tolerance 1e-4 mean(var), k-means++ seeding with
//                                                         2 + ln M candidates per centre, Lloyd until the labels stop
//                                                         changing or the centres move less than the tolerance
//   first M-step from the hard assignments, then EM       until |change of the mean log-likelihood| < tol or max_iter
//
// The random numbers are numpy's: RandomState(seed) is MT19937 seeded by init_genrand, random_sample() its 53-bit
// uniform; the seeding consumes one for the first centre (through the cumulative distribution numpy builds:
// np.full(n, 1/n).cumsum() / last, searchsorted side='right') and 2 + ln M per further centre.  With the same seed the
// result equals kwiiyatta_amd.converter.gmm_fit.GaussianMixtureHIP's and scikit-learn's (tests/test_gmm_fit.py).
// kwy_gmm_fit_comm_dev is the same control flow for the rows of SEVERAL ranks: the caller hands in its communicator
// as an all-reduce callback (RCCL: ncclAllReduce on the given stream), the library exchanges what the Python driver
// (kwiiyatta_amd/converter/gmm_fit.py) exchanges.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "kwy_internal.hpp"

namespace {

struct HostMT {           // numpy's legacy RandomState(seed) for an integer seed
  uint32_t key[624];
  int pos;
  explicit HostMT(uint32_t seed) {
    for (int i = 0; i < 624; ++i) {
      key[i] = seed;
      seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    pos = 624;
  }
  uint32_t next() {
    if (pos == 624) {
      auto tw = [](uint32_t u, uint32_t v, uint32_t far) {
        const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      };
      for (int k = 0; k < 227; ++k) key[k] = tw(key[k], key[k + 1], key[k + 397]);
      for (int k = 227; k < 623; ++k) key[k] = tw(key[k], key[k + 1], key[k - 227]);
      key[623] = tw(key[623], key[0], key[396]);
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  double random_sample() {
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};

template <typename T>
struct DevBuf {           // hipMalloc'ed for the duration of the fit
  T *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
};

}  // namespace

// distance of every row to the centre of its own label (empty-cluster relocation: sklearn takes the farthest rows)
__global__ __launch_bounds__(KWY_THREADS) void k_fit_own_dist(const double *__restrict__ Xc, const int *__restrict__ labels,
                                                             const double *__restrict__ centers, int64_t n, int D,
                                                             double *__restrict__ d) {
  const int64_t t = (int64_t)blockIdx.x * KWY_THREADS + threadIdx.x;
  if (t >= n) return;
  const double *x = Xc + t * D, *c = centers + (size_t)labels[t] * D;
  double s = 0.0;
  for (int k = 0; k < D; ++k) { const double v = x[k] - c[k]; s += v * v; }
  d[t] = s;
}

#define FIT_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return e_ == hipErrorOutOfMemory ? KWY_ENOMEM : KWY_EHIP;                    \
    }                                                                              \
  } while (0)

__global__ void k_fit_u64_to_f64(const unsigned long long *__restrict__ src, double *__restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = (double)src[0];
}

// The fit of this rank's rows; with a communicator the rows of all ranks (global row order: rank 0's rows, then rank
// 1's, ...).  Every exchange is an in-place SUM all-reduce of a small device buffer on the context's stream; every
// decision is taken on reduced numbers, and the random draws are the same on every rank, so all ranks return the same
// model -- the model of a one-rank fit of the concatenated rows, up to the rounding of the sums.
extern "C" int kwy_gmm_fit_comm_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int max_iter, double tol,
                                    double reg_covar, uint32_t seed, const kwy_comm *comm, double *weights,
                                    double *means, double *covs, int *n_iter_out, double *lower_bound_out,
                                    int *converged_out, int *kmeans_iter_out) {
  if (!ctx) return KWY_EINVAL;
  const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
  // What every rank can check about the GROUP comes first: without a usable communicator nothing can be agreed on.
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !comm->all_reduce_sum)) {
    ctx->err = "gmm_fit: bad communicator";
    return KWY_EINVAL;
  }
  // Everything that can fail on ONE rank only -- its arguments (an empty shard: n < 1), its allocations -- is collected
  // in rc_local and becomes the FIRST exchanged quantity below, so that all ranks leave with an error together
  // instead of one returning while its peers wait in the next all-reduce.  (A failure of the communicator's
  // callback itself, e.g. a Python exception in it, cannot be agreed on: it is fatal for the whole group.)
  int rc_local = KWY_OK;
  if (!X || !weights || !means || !covs || n < 1 || M < 1 || M > 256 || D < 1 || D > 160 || max_iter < 1) {
    ctx->err = "gmm_fit: bad argument (needs n >= 1 rows per rank, M <= 256, D <= 160)";
    rc_local = KWY_EINVAL;
  }
  const int n_trials = 2 + (int)log((double)(M > 0 ? M : 1));
  if (rc_local == KWY_OK && n_trials > 8) {
    ctx->err = "gmm_fit: more than 8 k-means++ candidates per centre (M > 403)";
    rc_local = KWY_EINVAL;
  }
  if (world == 1 && rc_local != KWY_OK) return rc_local;
  FIT_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nn = (size_t)(n > 0 ? n : 0);

  DevBuf<double> resp, ll, stats, sxx, dmeans, dweights, dcovs, Xc, xsq, closest, newd, pots, csums, total, shift2;
  DevBuf<double> centers, centers_new, cand, vals, cs, meanv, owndist, gath, lohi, pack;
  DevBuf<int64_t> pick;
  DevBuf<int32_t> pick32, labels;
  DevBuf<unsigned long long> changed;
  DevBuf<int> status;
  const size_t nchunks = rc_local == KWY_OK ? (size_t)kwy_km_chunks(n) : 1;
  const size_t nstats = (size_t)(M > 0 ? M : 1) * ((D > 0 ? D : 1) + 1);
  FIT_HIP(gath.alloc((size_t)world + 2));           // (the buffer of the agreement itself: its failure is fatal)
  auto setup = [&]() -> int {
    FIT_HIP(resp.alloc(nn * M)); FIT_HIP(ll.alloc((nn + 255) / 256)); FIT_HIP(stats.alloc(nstats + 1));
    FIT_HIP(sxx.alloc((size_t)M * D * D)); FIT_HIP(dmeans.alloc((size_t)M * D)); FIT_HIP(dweights.alloc(M));
    FIT_HIP(dcovs.alloc((size_t)M * D * D)); FIT_HIP(Xc.alloc(nn * D)); FIT_HIP(xsq.alloc(nn)); FIT_HIP(closest.alloc(nn));
    FIT_HIP(newd.alloc(8 * nn)); FIT_HIP(pots.alloc(8)); FIT_HIP(csums.alloc(nchunks)); FIT_HIP(total.alloc(1));
    FIT_HIP(shift2.alloc(M)); FIT_HIP(centers.alloc((size_t)M * D)); FIT_HIP(centers_new.alloc((size_t)M * D));
    FIT_HIP(cand.alloc(8 * (size_t)D)); FIT_HIP(vals.alloc(8)); FIT_HIP(cs.alloc(2 * (size_t)D)); FIT_HIP(meanv.alloc(D));
    FIT_HIP(pick.alloc(8)); FIT_HIP(pick32.alloc(8)); FIT_HIP(labels.alloc(nn)); FIT_HIP(changed.alloc(1));
    FIT_HIP(status.alloc(4)); FIT_HIP(lohi.alloc(2));
    FIT_HIP(hipMemsetAsync(status.p, 0, sizeof(int) * 4, st));
    FIT_HIP(hipMemsetAsync(labels.p, 0xff, sizeof(int32_t) * nn, st));     // -1: every label "changes" in the first pass
    return KWY_OK;
  };
  if (rc_local == KWY_OK) rc_local = setup();

  auto d2h = [&](void *dst, const void *src, size_t bytes) -> int {
    FIT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    FIT_HIP(hipStreamSynchronize(st));
    return KWY_OK;
  };
  auto h2d = [&](void *dst, const void *src, size_t bytes) -> int {
    FIT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    FIT_HIP(hipStreamSynchronize(st));          // the source is a local host buffer
    return KWY_OK;
  };
  auto all_reduce = [&](double *buf, size_t count) -> int {
    if (world == 1) return KWY_OK;
    if (comm->all_reduce_sum(comm->user, buf, (int64_t)count, (void *)st) != 0) {
      ctx->err = "gmm_fit: the communicator's all-reduce failed";
      return KWY_EHIP;
    }
    return KWY_OK;
  };
  // one double per rank -> all of them on the host, in rank order
  std::vector<double> hgath(world);
  auto all_gather = [&](const double *dev_scalar) -> int {
    FIT_HIP(hipMemsetAsync(gath.p, 0, sizeof(double) * world, st));
    FIT_HIP(hipMemcpyAsync(gath.p + rank, dev_scalar, sizeof(double), hipMemcpyDeviceToDevice, st));
    KWY_TRY(all_reduce(gath.p, world));
    return d2h(hgath.data(), gath.p, sizeof(double) * world);
  };

  // ---------------------------------------------------------------- the rows of all ranks, and whether all are ready
  {
    // slot r: the rows of rank r; slot `world`: the number of ranks whose set-up failed
    std::vector<double> mine((size_t)world + 1, 0.0);
    mine[rank] = (double)(n > 0 ? n : 0);
    mine[world] = rc_local != KWY_OK ? 1.0 : 0.0;
    const std::string local_err = ctx->err;
    KWY_TRY(h2d(gath.p, mine.data(), sizeof(double) * (world + 1)));
    KWY_TRY(all_reduce(gath.p, (size_t)world + 1));
    hgath.resize((size_t)world + 1);
    KWY_TRY(d2h(hgath.data(), gath.p, sizeof(double) * (world + 1)));
    if (hgath[world] > 0.0) {
      if (rc_local != KWY_OK) { ctx->err = local_err; return rc_local; }
      ctx->err = "gmm_fit: another rank failed its set-up (bad arguments, an empty shard or an allocation); no rank fits";
      return KWY_EINVAL;
    }
  }
  int64_t n_total = 0, row0 = 0;
  for (int r = 0; r < world; ++r) { const int64_t c = (int64_t)llround(hgath[r]); if (r < rank) row0 += c; n_total += c; }
  if (n_total < M) { ctx->err = "gmm_fit: fewer rows than mixture components"; return KWY_EINVAL; }
  const double nt = (double)n_total;

  // ---------------------------------------------------------------- KMeans.fit: centring, tolerance
  std::vector<double> hcs(2 * (size_t)D), hmean(D);
  KWY_TRY(kwy_km_colstats_dev(ctx, X, n, D, nullptr, cs.p));
  KWY_TRY(all_reduce(cs.p, 2 * (size_t)D));
  KWY_TRY(d2h(hcs.data(), cs.p, sizeof(double) * 2 * D));
  for (int k = 0; k < D; ++k) hmean[k] = hcs[k] / nt;
  KWY_TRY(h2d(meanv.p, hmean.data(), sizeof(double) * D));
  KWY_TRY(kwy_km_center_dev(ctx, X, n, D, meanv.p, Xc.p, xsq.p));
  KWY_TRY(kwy_km_colstats_dev(ctx, X, n, D, meanv.p, cs.p));
  KWY_TRY(all_reduce(cs.p, 2 * (size_t)D));
  KWY_TRY(d2h(hcs.data(), cs.p, sizeof(double) * 2 * D));
  double var_mean = 0.0;
  for (int k = 0; k < D; ++k) { const double m1 = hcs[k] / nt; var_mean += hcs[D + k] / nt - m1 * m1; }
  const double abs_tol = var_mean / D * 1e-4;

  // ---------------------------------------------------------------- _kmeans_plusplus with numpy's draws
  HostMT rs(seed);
  int64_t first_id;
  {
    // random_state.choice(n) of scikit-learn >= 1.3 goes through the cumulative distribution of uniform weights
    const double r = rs.random_sample(), w = 1.0 / nt;
    std::vector<double> cdf((size_t)n_total);
    double run = 0.0;
    for (size_t i = 0; i < cdf.size(); ++i) { run += w; cdf[i] = run; }
    const double last = cdf.back();
    for (size_t i = 0; i < cdf.size(); ++i) cdf[i] /= last;
    first_id = (int64_t)(std::upper_bound(cdf.begin(), cdf.end(), r) - cdf.begin());
    if (first_id >= n_total) first_id = n_total - 1;
  }
  std::vector<double> u((size_t)(M > 1 ? M - 1 : 0) * n_trials);
  for (double &v : u) v = 0.0 + (1.0 - 0.0) * rs.random_sample();      // uniform(size=(M - 1, n_trials))

  // rows of the global matrix by LOCAL index (-1: another rank's) into cand[0 .. L): every rank gets all of them
  auto fetch_rows = [&](const int64_t *idx, int L) -> int {
    FIT_HIP(hipMemsetAsync(cand.p, 0, sizeof(double) * (size_t)L * D, st));
    for (int q = 0; q < L; ++q)
      if (idx[q] >= 0 && idx[q] < n)
        FIT_HIP(hipMemcpyAsync(cand.p + (size_t)q * D, Xc.p + (size_t)idx[q] * D, sizeof(double) * D, hipMemcpyDeviceToDevice, st));
    return all_reduce(cand.p, (size_t)L * D);
  };
  {
    const int64_t local = first_id - row0;
    KWY_TRY(fetch_rows(&local, 1));
    FIT_HIP(hipMemcpyAsync(centers.p, cand.p, sizeof(double) * D, hipMemcpyDeviceToDevice, st));
  }
  KWY_TRY(kwy_km_pp_dist_dev(ctx, Xc.p, xsq.p, n, D, centers.p, 1, nullptr, newd.p, pots.p));
  FIT_HIP(hipMemcpyAsync(closest.p, newd.p, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
  double hpots[8];
  KWY_TRY(all_reduce(pots.p, 1));
  KWY_TRY(d2h(hpots, pots.p, sizeof(double)));
  double current_pot = hpots[0];
  for (int c = 1; c < M; ++c) {
    KWY_TRY(kwy_km_pp_total_dev(ctx, closest.p, n, csums.p, total.p));
    double hvals[8];
    for (int q = 0; q < n_trials; ++q) hvals[q] = u[(size_t)(c - 1) * n_trials + q] * current_pot;
    KWY_TRY(h2d(vals.p, hvals, sizeof(double) * n_trials));
    if (world > 1) {
      // shard r owns the values in (bounds[r], bounds[r + 1]]: one cumulative sum of the gathered totals, the same
      // numbers on every rank, so that a value on a boundary has exactly one owner
      KWY_TRY(all_gather(total.p));
      double b[2] = {0.0, 0.0}, run = 0.0;
      for (int r = 0; r < world; ++r) { if (r == rank) b[0] = run; run += hgath[r]; if (r == rank) b[1] = run; }
      KWY_TRY(h2d(lohi.p, b, sizeof(double) * 2));
      KWY_TRY(kwy_km_pp_pick_dev(ctx, closest.p, n, csums.p, lohi.p, lohi.p + 1, vals.p, n_trials, rank == 0 ? 1 : 0,
                                 rank == world - 1 ? 1 : 0, pick.p));
    } else {
      KWY_TRY(kwy_km_pp_pick_dev(ctx, closest.p, n, csums.p, nullptr, nullptr, vals.p, n_trials, 1, 1, pick.p));
    }
    int64_t hpick[8];
    KWY_TRY(d2h(hpick, pick.p, sizeof(int64_t) * n_trials));
    if (world == 1)
      for (int q = 0; q < n_trials; ++q) hpick[q] = hpick[q] < 0 ? 0 : (hpick[q] >= n ? n - 1 : hpick[q]);
    KWY_TRY(fetch_rows(hpick, n_trials));
    KWY_TRY(kwy_km_pp_dist_dev(ctx, Xc.p, xsq.p, n, D, cand.p, n_trials, closest.p, newd.p, pots.p));
    KWY_TRY(all_reduce(pots.p, n_trials));
    KWY_TRY(d2h(hpots, pots.p, sizeof(double) * n_trials));
    int best = 0;
    for (int q = 1; q < n_trials; ++q) if (hpots[q] < hpots[best]) best = q;      // np.argmin: first minimum
    FIT_HIP(hipMemcpyAsync(closest.p, newd.p + (size_t)best * nn, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    FIT_HIP(hipMemcpyAsync(centers.p + (size_t)c * D, cand.p + (size_t)best * D, sizeof(double) * D, hipMemcpyDeviceToDevice, st));
    current_pot = hpots[best];
  }

  // ---------------------------------------------------------------- _kmeans_single_lloyd
  std::vector<double> hstats(nstats + 1), hshift(M);
  double *cur = centers.p, *nxt = centers_new.p;
  bool strict = false;
  int km_iter = 0;
  for (km_iter = 1; km_iter <= 300; ++km_iter) {
    KWY_TRY(kwy_km_assign_dev(ctx, Xc.p, n, D, cur, M, labels.p, resp.p, changed.p));
    KWY_TRY(kwy_gmm_em_sums_dev(ctx, Xc.p, n, D, M, resp.p, stats.p));
    hipLaunchKernelGGL(k_fit_u64_to_f64, dim3(1), dim3(64), 0, st, changed.p, stats.p + nstats);   // (reduced with the sums)
    KWY_TRY(all_reduce(stats.p, nstats + 1));
    KWY_TRY(d2h(hstats.data(), stats.p, sizeof(double) * hstats.size()));
    std::vector<int> empty;
    for (int j = 0; j < M; ++j) if (hstats[(size_t)j * (D + 1)] == 0.0) empty.push_back(j);
    if (!empty.empty()) {
      // _relocate_empty_clusters_dense: every empty cluster takes one of the rows farthest from their own centre
      // (largest first; scikit-learn's np.argpartition leaves the order among the k farthest unspecified -- with two
      // or more empty clusters in one iteration its pairing may differ), which leaves its old cluster.  Every rank
      // offers its k farthest rows {distance, label, row}; all ranks pick the same k of them.
      if (!owndist.p) FIT_HIP(owndist.alloc(nn));
      hipLaunchKernelGGL(k_fit_own_dist, dim3((unsigned)((nn + KWY_THREADS - 1) / KWY_THREADS)), dim3(KWY_THREADS), 0, st,
                         Xc.p, labels.p, cur, n, D, owndist.p);
      std::vector<double> hd(nn);
      KWY_TRY(d2h(hd.data(), owndist.p, sizeof(double) * nn));
      std::vector<int64_t> order(nn);
      for (size_t i = 0; i < nn; ++i) order[i] = (int64_t)i;
      const size_t kk = empty.size(), k = std::min(kk, nn);
      std::partial_sort(order.begin(), order.begin() + k, order.end(), [&](int64_t a, int64_t b) {
        return hd[a] > hd[b] || (hd[a] == hd[b] && a < b);
      });
      const size_t rec = (size_t)D + 2;               // {distance, old label, row}
      std::vector<double> hpack((size_t)world * kk * rec, 0.0);
      for (size_t e = 0; e < kk; ++e) {
        double *o = hpack.data() + ((size_t)rank * kk + e) * rec;
        if (e >= k) { o[0] = -1.0; continue; }
        int32_t old;
        o[0] = hd[order[e]];
        KWY_TRY(d2h(&old, labels.p + order[e], sizeof(int32_t)));
        o[1] = (double)old;
        KWY_TRY(d2h(o + 2, Xc.p + (size_t)order[e] * D, sizeof(double) * D));
      }
      if (world > 1) {
        for (int r = 0; r < world; ++r)
          if (r != rank) for (size_t e = 0; e < kk; ++e) hpack[((size_t)r * kk + e) * rec] = 0.0;
        if (pack.p) { (void)hipFree(pack.p); pack.p = nullptr; }
        FIT_HIP(pack.alloc(hpack.size()));
        KWY_TRY(h2d(pack.p, hpack.data(), sizeof(double) * hpack.size()));
        KWY_TRY(all_reduce(pack.p, hpack.size()));
        KWY_TRY(d2h(hpack.data(), pack.p, sizeof(double) * hpack.size()));
      }
      std::vector<size_t> slot((size_t)world * kk);
      for (size_t i = 0; i < slot.size(); ++i) slot[i] = i;
      std::stable_sort(slot.begin(), slot.end(), [&](size_t a, size_t b) { return hpack[a * rec] > hpack[b * rec]; });
      for (size_t e = 0; e < kk && e < slot.size(); ++e) {
        const double *o = hpack.data() + slot[e] * rec;
        if (o[0] < 0.0) break;
        const int old = (int)o[1];
        double *so = hstats.data() + (size_t)old * (D + 1), *sj = hstats.data() + (size_t)empty[e] * (D + 1);
        so[0] -= 1.0;
        sj[0] = 1.0;
        for (int q = 0; q < D; ++q) { so[1 + q] -= o[2 + q]; sj[1 + q] = o[2 + q]; }
      }
      KWY_TRY(h2d(stats.p, hstats.data(), sizeof(double) * nstats));
    }
    KWY_TRY(kwy_km_update_dev(ctx, stats.p, cur, M, D, nxt, shift2.p));
    std::swap(cur, nxt);
    KWY_TRY(d2h(hshift.data(), shift2.p, sizeof(double) * M));
    const double hchanged = hstats[nstats];
    double shift_tot = 0.0;
    for (int j = 0; j < M; ++j) shift_tot += hshift[j];
    if (hchanged == 0.0) { strict = true; break; }
    if (shift_tot <= abs_tol) break;
  }
  if (km_iter > 300) km_iter = 300;
  if (!strict) KWY_TRY(kwy_km_assign_dev(ctx, Xc.p, n, D, cur, M, labels.p, resp.p, changed.p));   // labels of the final centres
  if (kmeans_iter_out) *kmeans_iter_out = km_iter;

  // ---------------------------------------------------------------- EM (BaseMixture.fit_predict)
  auto m_step = [&]() -> int {
    KWY_TRY(kwy_gmm_em_sums_dev(ctx, X, n, D, M, resp.p, stats.p));
    KWY_TRY(all_reduce(stats.p, nstats));
    KWY_TRY(kwy_gmm_em_means_dev(ctx, stats.p, D, M, dmeans.p));
    KWY_TRY(kwy_gmm_em_cov_stats_dev(ctx, X, n, D, M, resp.p, dmeans.p, stats.p, sxx.p));
    KWY_TRY(all_reduce(sxx.p, (size_t)M * D * D));
    return kwy_gmm_em_finalize_dev(ctx, stats.p, sxx.p, D, M, reg_covar, dweights.p, dcovs.p);
  };
  KWY_TRY(m_step());                                     // initialisation from the hard assignments
  double lower_bound = -INFINITY;
  int converged = 0, it = 0;
  std::vector<double> hll((nn + 255) / 256);
  for (it = 1; it <= max_iter; ++it) {
    const double prev = lower_bound;
    KWY_TRY(kwy_gmm_em_estep_dev(ctx, X, n, D, M, dweights.p, dmeans.p, dcovs.p, resp.p, ll.p, status.p));
    int hstatus = 0;
    KWY_TRY(d2h(hll.data(), ll.p, sizeof(double) * hll.size()));
    KWY_TRY(d2h(&hstatus, status.p, sizeof(int)));
    double both[2] = {0.0, (double)(hstatus != 0)};
    for (double v : hll) both[0] += v;
    if (world > 1) {
      KWY_TRY(h2d(lohi.p, both, sizeof(double) * 2));
      KWY_TRY(all_reduce(lohi.p, 2));
      KWY_TRY(d2h(both, lohi.p, sizeof(double) * 2));
    }
    if (both[1] != 0.0) {
      ctx->err = "gmm_fit: some components have ill-defined empirical covariance; increase reg_covar";
      return KWY_ENUMERIC;
    }
    KWY_TRY(m_step());
    lower_bound = both[0] / nt;
    if (fabs(lower_bound - prev) < tol) { converged = 1; break; }
  }
  if (it > max_iter) it = max_iter;
  KWY_TRY(d2h(weights, dweights.p, sizeof(double) * M));
  KWY_TRY(d2h(means, dmeans.p, sizeof(double) * (size_t)M * D));
  KWY_TRY(d2h(covs, dcovs.p, sizeof(double) * (size_t)M * D * D));
  if (n_iter_out) *n_iter_out = it;
  if (lower_bound_out) *lower_bound_out = lower_bound;
  if (converged_out) *converged_out = converged;
  return KWY_OK;
}

extern "C" int kwy_gmm_fit_dev(kwy_ctx *ctx, const double *X, int64_t n, int D, int M, int max_iter, double tol,
                               double reg_covar, uint32_t seed, double *weights, double *means, double *covs,
                               int *n_iter_out, double *lower_bound_out, int *converged_out, int *kmeans_iter_out) {
  if (ctx && n < M) { ctx->err = "gmm_fit: bad argument (needs n >= M, M <= 256, D <= 160)"; return KWY_EINVAL; }
  return kwy_gmm_fit_comm_dev(ctx, X, n, D, M, max_iter, tol, reg_covar, seed, nullptr, weights, means, covs, n_iter_out,
                              lower_bound_out, converged_out, kmeans_iter_out);
}
