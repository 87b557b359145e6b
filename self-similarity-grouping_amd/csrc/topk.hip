// topk.hip -- K5: per-row top-(k1+1) ranking of the normalised original distance.
//
// Replaces reid/rerank.py:68-70:
//   original_dist = transpose(original_dist / max(original_dist, axis=0))
//   initial_rank  = argsort(original_dist)          (only columns < k1+1 are ever read)
// D is symmetric, so row i of the normalised matrix is half(D[i,:] / rowmax[i]); it is
// recomputed in registers and never written to HBM.  Order is the canonical
// (normalised half value, column index) order == numpy argsort(kind='stable').
//
// HBM-bound streaming kernel: one wave per row reads the row once (16 B per lane per
// load).  The wave keeps the 64 smallest (key,index) pairs sorted across its lanes; a raw
// 16-bit threshold prefilter rejects almost every element with one integer compare, so
// the float divide only runs for the ~K*ln(N/K) elements that can still enter the list.
#include "ssg_common.h"

namespace ssg {

// composite sort key: [63:48] normalised half bits | [47:16] column | [15:0] raw D bits
__device__ __forceinline__ uint64_t make_comp(hbits key, int col, hbits raw) {
  return ((uint64_t)key << 48) | ((uint64_t)(uint32_t)col << 16) | (uint64_t)raw;
}

// 64-bit lane exchange helpers: broadcast from a wave-uniform lane via v_readlane (SALU path, no
// LDS permute), shift-by-one via DPP wave_shr:1.
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shr1_64(uint64_t v) {   // lane l gets lane l-1 (lane 0 keeps its own)
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void topk_rank_kernel(const hbits* __restrict__ D, const unsigned* __restrict__ rowmax, int N,
                                                        int nrows, int K, int32_t* __restrict__ rank) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= nrows) return;
  const int lane = lane_id();
  const float fmx = h2f((hbits)rowmax[row]);
  const bool degenerate = !(fmx > 0.f) || fmx > 65504.f;   // all-zero / NaN row: keys are all NaN, keep every column eligible
  const int64_t total = (int64_t)nrows * N;
  const int64_t base = (int64_t)row * N;
  const int64_t al = base & ~(int64_t)7;     // 16-byte aligned element offset at/below the row start
  const int first = (int)(base - al);
  const int nchunks = (first + N + 511) / 512;

  uint64_t mine = ~0ULL;       // lane r holds the r-th smallest composite seen so far
  uint64_t tau = ~0ULL;        // composite at lane K-1 (the current K-th smallest)
  unsigned raw_hi = 0xffffu;   // raw values above this cannot beat tau (superset prefilter)

  auto load_chunk = [&](int c) -> uint4 {
    const int64_t off = al + (int64_t)c * 512 + lane * 8;
    uint4 x = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (off + 8 <= total) x = *reinterpret_cast<const uint4*>(D + off);
    else if (off < total) {
      unsigned short t[8];
#pragma unroll
      for (int e = 0; e < 8; e++) t[e] = (off + e < total) ? D[off + e] : (unsigned short)0xffff;
      x = make_uint4(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16));
    }
    return x;
  };

  uint4 nxt = load_chunk(0), nxt2 = nchunks > 1 ? load_chunk(1) : nxt;
  for (int c = 0; c < nchunks; c++) {
    const uint4 cur = nxt;
    nxt = nxt2;
    if (c + 2 < nchunks) nxt2 = load_chunk(c + 2);
    const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
    const int j0 = c * 512 + lane * 8 - first;   // row-relative column of element 0
    const bool interior = (c > 0) && (c + 1 < nchunks);   // every lane's 8 columns are inside the row
    // per-lane bitmask of elements that pass the raw prefilter
    unsigned pm = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const unsigned r = (w[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
      const int j = j0 + e;
      if ((interior || (j >= 0 && j < N)) && r <= raw_hi) pm |= 1u << e;
    }
    // drain the masks: one element per lane per round (typically a single round with 1-2 lanes)
    while (__any(pm != 0)) {
      uint64_t comp = ~0ULL;
      if (pm) {
        const int e = __ffs((int)pm) - 1;
        pm &= pm - 1;
        const unsigned wsel = e < 2 ? w[0] : e < 4 ? w[1] : e < 6 ? w[2] : w[3];
        const hbits r = (hbits)((wsel >> ((e & 1) * 16)) & 0xffffu);
        if (r <= raw_hi) comp = make_comp(f2h(h2f(r) / fmx), j0 + e, r);   // raw_hi may have tightened since pm was built
      }
      bool cand = comp < tau;
      uint64_t mask = __ballot(cand);
      while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        const uint64_t cc = bcast64(comp, src);
        // sorted insert of cc into the 64-lane list
        const bool lt = mine < cc;
        const uint64_t up = shr1_64(mine);
        const int pos = __popcll(__ballot(lt));
        mine = lt ? mine : (lane == pos ? cc : up);
        tau = bcast64(mine, K - 1);
        if (tau != ~0ULL && !degenerate) {
          const hbits ktau = (hbits)(tau >> 48);
          unsigned rr = (unsigned)(tau & 0xffffu);
          if (rr >= 0x7c00u) raw_hi = 0xffffu;
          else if (ktau >= 0x0400u) {
            // normal-range key: the two half grids have a spacing ratio in (1/2, 2], so at most 3
            // neighbouring raw values share a key; +4 keeps the prefilter a superset and the
            // exact (key, column) compare decides.
            raw_hi = rr + 4u < 0x7c00u ? rr + 4u : 0x7c00u;
          } else {
            // subnormal key: arbitrarily many raw values can collapse onto it -> exact bound
            while (rr < 0x7c00u && f2h(h2f((hbits)(rr + 1)) / fmx) == ktau) rr++;
            raw_hi = rr;
          }
        }
        if (lane == src) cand = false;
        cand = cand && (comp < tau);
        mask = __ballot(cand);
      }
    }
  }
  if (lane < K) rank[(int64_t)row * K + lane] = (int32_t)((mine >> 16) & 0xffffffffULL);
}

}  // namespace ssg

using namespace ssg;

extern "C" int ssg_topk_rank(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || K <= 0 || K > 64 || K > N) {
    ssg_set_error("ssg_topk_rank: need 0 < K <= min(64, N) (K=%d N=%d)", K, N);
    return SSG_ERR_INVALID;
  }
  hipLaunchKernelGGL(topk_rank_kernel, dim3((nrows + 3) / 4), dim3(256), 0, stream, D, rowmax, N, nrows, K, rank);
  SSG_LAUNCH_CHECK("topk_rank_kernel");
  return SSG_OK;
}
