// ranking.hip -- retrieval metrics of the per-iteration evaluation.
//
// Replaces reid/evaluators.py:88-129 evaluate_all -> reid/evaluation_metrics/ranking.py:18-79 cmc (the
// 'market1501' configuration: first_match_break, optional separate_camera_set) and :82-115 mean_ap
// (sklearn average_precision_score per query) on a query x gallery float32 distance block.
//
// The reference argsorts every row (m x n log n on one core).  Neither metric needs the order of the
// non-matching entries: with the true matches T of a query sorted by (distance, index),
//   rank of the first match     = #{valid j ordered before it}
//   precision at a threshold t  = #{matches <= t} / #{valid <= t},  recall = #{matches <= t} / |T|
// so one workgroup per query gathers T into LDS, sorts it (|T| is tens), and makes ONE more pass over the
// row, binary-searching each valid distance into T and counting per bucket.  AP is then the sum over the
// distinct match distances of (recall step) x precision, exactly sklearn's step-wise definition (equal
// scores form one threshold; float64 quotients like sklearn's, summation order differs -> 1e-12 agreement).
#include "ssg_common.h"

namespace ssg {

constexpr int RK_CAP = 2048;   // true matches of one query held in LDS

__global__ __launch_bounds__(256) void rank_metrics_kernel(const float* __restrict__ dist, int64_t ld, int n, const int* __restrict__ qid,
                                                           const int* __restrict__ qcam, const int* __restrict__ gid, const int* __restrict__ gcam,
                                                           int separate_cams, int* __restrict__ first_rank, double* __restrict__ ap,
                                                           int* __restrict__ overflow, int* __restrict__ nm_before, int* __restrict__ nmatch,
                                                           int nm_cap) {
  __shared__ float pd[RK_CAP], sd[RK_CAP];
  __shared__ int pj[RK_CAP], sj[RK_CAP];
  __shared__ unsigned hist[RK_CAP + 1];
  __shared__ unsigned hist2[RK_CAP + 1];      // the same by (distance, index) order: bin b = valid entries between match b-1 and match b (incl. match b)
  __shared__ int npos, before;
  const int q = (int)blockIdx.x, tid = (int)threadIdx.x;
  const float* row = dist + (int64_t)q * ld;
  const int myid = qid[q], mycam = qcam[q];
  if (tid == 0) { npos = 0; before = 0; }
  for (int t = tid; t <= RK_CAP; t += 256) { hist[t] = 0; hist2[t] = 0; }
  __syncthreads();
  // valid gallery entry: different identity or different camera (ranking.py:47-48,105-106); the 'separate camera set'
  // protocol drops the whole query camera (:49-51)
  auto is_valid = [&](int j) { const bool diffcam = gcam[j] != mycam; return separate_cams ? diffcam : (gid[j] != myid || diffcam); };
  for (int j = tid; j < n; j += 256)
    if (gid[j] == myid && is_valid(j)) {
      const int k = atomicAdd(&npos, 1);
      if (k < RK_CAP) { pd[k] = row[j]; pj[k] = j; }
    }
  __syncthreads();
  const int P = npos;
  if (P == 0 || P > RK_CAP) {   // no valid true match: the reference skips the query (ranking.py:51,110)
    if (tid == 0) { first_rank[q] = -1; ap[q] = __longlong_as_double(0x7ff8000000000000LL); if (P > RK_CAP) atomicAdd(overflow, 1); if (nmatch) nmatch[q] = 0; }
    return;
  }
  // rank sort of the matches by (distance, index)
  for (int t = tid; t < P; t += 256) {
    const float x = pd[t]; const int jx = pj[t];
    int r = 0;
    for (int u = 0; u < P; u++) r += (pd[u] < x || (pd[u] == x && pj[u] < jx)) ? 1 : 0;
    sd[r] = x; sj[r] = jx;
  }
  __syncthreads();
  const float d0 = sd[0]; const int j0 = sj[0];
  int mine = 0;
  for (int j = tid; j < n; j += 256) {
    if (!is_valid(j)) continue;
    const float x = row[j];
    int lo = 0, hi = P;                      // lower bound: number of matches with distance < x
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sd[mid] < x) lo = mid + 1; else hi = mid; }
    atomicAdd(&hist[lo], 1u);
    mine += (x < d0 || (x == d0 && j < j0)) ? 1 : 0;
    if (nm_before) {                         // number of matches ordered before (x, j) in (distance, index) order
      int hi2 = lo;                          // matches with distance < x are before; among those with distance == x, the smaller indices
      while (hi2 < P && sd[hi2] == x && sj[hi2] < j) hi2++;
      atomicAdd(&hist2[hi2], 1u);
    }
  }
  atomicAdd(&before, mine);
  __syncthreads();
  if (tid == 0) {
    first_rank[q] = before;
    double acc = 0.0, prev_recall = 0.0;
    unsigned long long n_le = 0;
    int b = 0;
    for (int s = 0; s < P;) {
      int e = s + 1;
      while (e < P && sd[e] == sd[s]) e++;
      for (; b <= s; b++) n_le += hist[b];   // valid entries with distance <= sd[s]
      const double recall = (double)e / (double)P, precision = (double)e / (double)n_le;
      acc += (recall - prev_recall) * precision;
      prev_recall = recall;
      s = e;
    }
    ap[q] = acc;
  }
  if (nm_before) {
    // all-shots CMC (ranking.py:62-75 without first_match_break): match s sits at valid-list rank k_s = sum_{b<=s} hist2[b] - 1,
    // the reference credits bin k_s - s = number of valid NON-matching entries ordered before it
    __syncthreads();
    if (tid == 0) {
      unsigned long long run = 0;
      const int lim = P < nm_cap ? P : nm_cap;
      for (int sidx = 0; sidx < lim; sidx++) { run += hist2[sidx]; nm_before[(int64_t)q * nm_cap + sidx] = (int)(run - 1 - (unsigned long long)sidx); }
      nmatch[q] = P;
    }
  }
}

}  // namespace ssg

using namespace ssg;

// first_rank[q] = number of valid gallery entries ordered before query q's first true match in (distance, index)
// order (-1: no valid true match), ap[q] = average precision (NaN likewise).  dist [m, ld] float32, ids / cams int32.
// *overflow counts queries with more than 2048 true matches (reported as invalid; the caller raises).
static int rank_metrics_impl(const float* dist, int m, int n, int64_t ld, const int32_t* qid, const int32_t* qcam, const int32_t* gid,
                             const int32_t* gcam, int separate_cams, int32_t* first_rank, double* ap, int32_t* overflow, int32_t* nm_before,
                             int32_t* nmatch, int nm_cap, hipStream_t stream) {
  if (m <= 0 || n <= 0 || ld < n) { ssg_set_error("ssg_rank_metrics: bad shape m=%d n=%d ld=%lld", m, n, (long long)ld); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(rank_metrics_kernel, dim3(m), dim3(256), 0, stream, dist, ld, n, qid, qcam, gid, gcam, separate_cams, first_rank, ap, overflow,
                     nm_before, nmatch, nm_cap);
  SSG_LAUNCH_CHECK("rank_metrics_kernel");
  return SSG_OK;
}
extern "C" int ssg_rank_metrics(const float* dist, int m, int n, int64_t ld, const int32_t* qid, const int32_t* qcam, const int32_t* gid,
                                const int32_t* gcam, int separate_cams, int32_t* first_rank, double* ap, int32_t* overflow, hipStream_t stream) {
  return rank_metrics_impl(dist, m, n, ld, qid, qcam, gid, gcam, separate_cams, first_rank, ap, overflow, nullptr, nullptr, 0, stream);
}
// the same plus, per query, the all-shots CMC bins: nmatch[q] = number of valid true matches, nm_before[q, s] (s < min(nmatch, nm_cap)) =
// number of valid non-matching gallery entries ordered before the s-th match (ranking.py:62-75 with first_match_break=False)
extern "C" int ssg_rank_metrics_all(const float* dist, int m, int n, int64_t ld, const int32_t* qid, const int32_t* qcam, const int32_t* gid,
                                    const int32_t* gcam, int separate_cams, int32_t* first_rank, double* ap, int32_t* overflow, int32_t* nm_before,
                                    int32_t* nmatch, int nm_cap, hipStream_t stream) {
  if (!nm_before || !nmatch || nm_cap <= 0) { ssg_set_error("ssg_rank_metrics_all: need nm_before / nmatch / nm_cap"); return SSG_ERR_INVALID; }
  return rank_metrics_impl(dist, m, n, ld, qid, qcam, gid, gcam, separate_cams, first_rank, ap, overflow, nm_before, nmatch, nm_cap, stream);
}
