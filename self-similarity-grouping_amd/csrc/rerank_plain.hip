// rerank_plain.hip -- the kNN-set Jaccard re-ranking variant.
//
// Replaces reid/rerank_plain.py:125-178 re_ranking after the two steps it shares with rerank.py (source term :130-143 and
// half original distance :145-161 -- csrc/gram_i8.hip, pairwise.hip, conv.hip):
//   knn_bool[i] = { j != i : D[i,j] <= k-th smallest value of row i }                (:165-170, np.partition)
//   jaccard     = cdist(knn_bool, knn_bool, 'jaccard') = |A xor B| / |A or B| in float64, 0 for two empty sets -> half (:173)
//   final       = jaccard*(1-lambda) + source_dist*lambda                            (:175)
// As for rerank.py only J' = half(jaccard * half(1-lambda)) is written (2 bytes per entry); the eps rule and DBSCAN rebuild
// the float64 final_dist from J' and the source vector v exactly (final_dist_value()).
//
// The sets are tiny (k-1 members plus ties), so |A_i and A_k| is non-zero for a few hundred k per row: the same
// inverted-index walk as the Jaccard rows of rerank.py (csrc/jaccard.hip) with integer counts in LDS, a streaming
// constant fill (disjoint sets: distance 1) and a sparse patch of the touched columns.
#include "ssg_common.h"

namespace ssg {

// One wave per row: members of A_i in ascending column order.  thr = raw half bits of the k-th smallest entry (taken from
// the top-k list: D >= 0, so the bit order is the value order).  a_val = half(1) so that ssg_invert_index can be reused.
__global__ __launch_bounds__(256) void knn_set_kernel(const hbits* __restrict__ D, const int32_t* __restrict__ rank, int N, int row0, int nrows,
                                                      int K, int cap, int32_t* __restrict__ a_idx, hbits* __restrict__ a_val,
                                                      int32_t* __restrict__ a_nnz, int32_t* __restrict__ overflow) {
  const int il = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (il >= nrows) return;
  const int lane = lane_id(), gi = row0 + il;
  const hbits* row = D + (int64_t)il * N;
  const unsigned thr = row[rank[(int64_t)il * K + (K - 1)]];
  int n = 0;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    const bool hit = j < N && j != gi && (unsigned)row[j] <= thr;
    const uint64_t m = __ballot(hit);
    if (hit) {
      const int w = n + __popcll(m & lanemask_lt());
      if (w < cap) { a_idx[(int64_t)il * cap + w] = j; a_val[(int64_t)il * cap + w] = H_ONE; }
    }
    n += __popcll(m);
  }
  if (lane == 0) { a_nnz[il] = n < cap ? n : cap; if (n > cap) atomicAdd(overflow, 1); }
}

constexpr int PCHUNK = 32768;   // columns per LDS pass (64 KiB of uint16 counters)
constexpr int PTCAP = 3072;     // touched-column list per wave

__device__ __forceinline__ hbits set_jaccard_scaled(int ni, int nk, int c, hbits om) {
  const int denom = ni + nk - c, num = denom - c;          // |A or B|, |A xor B|
  const double d = denom == 0 ? 0.0 : (double)num / (double)denom;
  return h_mul(d2h(d), om);                                // half(jaccard) * half(1-lambda) in half
}

// Persistent waves, one row at a time: cnt[k] = |A_i and A_k| for the columns of one chunk.
__global__ __launch_bounds__(64) void set_jaccard_rows_kernel(const int32_t* __restrict__ a_nnz, int capA, const int32_t* __restrict__ a_idx,
                                                              const int64_t* __restrict__ colptr, const int32_t* __restrict__ inv_row, int N,
                                                              int row0, int nrows, hbits om, hbits* __restrict__ Jp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int cwmax = N < PCHUNK ? N : PCHUNK;
  unsigned short* cnt = reinterpret_cast<unsigned short*>(smem);
  unsigned short* touched = reinterpret_cast<unsigned short*>(smem + (((size_t)cwmax * 2 + 15) & ~(size_t)15) + 16);
  for (int x = lane * 8; x < cwmax; x += 512) *reinterpret_cast<uint4*>(cnt + x) = make_uint4(0, 0, 0, 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint64_t lt = lanemask_lt();
  const hbits jp1 = set_jaccard_scaled(1, 1, 0, om);       // disjoint, not both empty: distance 1

  for (int il = (int)blockIdx.x; il < nrows; il += (int)gridDim.x) {
    const int i = row0 + il, ni = a_nnz[i];
    for (int cbase = 0; cbase < N; cbase += PCHUNK) {
      const int cw = (N - cbase) < PCHUNK ? (N - cbase) : PCHUNK;
      int ntouched = 0;
      for (int p = 0; p < ni; p++) {
        const int j = a_idx[(int64_t)i * capA + p];
        const int64_t e0 = colptr[j], e1 = colptr[j + 1];    // rows k with j in A_k (distinct k: no two lanes hit one counter)
        for (int64_t eb = e0; eb < e1; eb += 64) {
          const int64_t e = eb + lane;
          const int kk = (e < e1 ? inv_row[e] : -1) - cbase;
          const bool hit = kk >= 0 && kk < cw;
          bool first = false;
          if (hit) { const unsigned short old = cnt[kk]; first = old == 0; cnt[kk] = (unsigned short)(old + 1); }
          const uint64_t fm = __ballot(first);
          if (first) { const int w = ntouched + __popcll(fm & lt); if (w < PTCAP) touched[w] = (unsigned short)kk; }
          ntouched += __popcll(fm);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      hbits* out = Jp + (int64_t)il * N + cbase;
      if (ni > 0 && ntouched <= PTCAP) {
        // (1) stream "distance 1" over the chunk, (2) wait for the stores, (3) patch the columns that share a neighbour
        const unsigned c2 = (unsigned)jp1 | ((unsigned)jp1 << 16);
        const int64_t eoff = (int64_t)il * N + cbase;
        const int head = (int)((8 - (eoff & 7)) & 7);
        for (int x = lane; x < head && x < cw; x += 64) out[x] = jp1;
        const int nvec = cw > head ? (cw - head) / 8 : 0;
        for (int q = lane; q < nvec; q += 64) *reinterpret_cast<uint4*>(out + head + q * 8) = make_uint4(c2, c2, c2, c2);
        for (int x = head + nvec * 8 + lane; x < cw; x += 64) out[x] = jp1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int q = lane; q < ntouched; q += 64) {
          const int kk = touched[q];
          out[kk] = set_jaccard_scaled(ni, a_nnz[cbase + kk], cnt[kk], om);
          cnt[kk] = 0;
        }
      } else {   // empty A_i (distance 0 to other empty sets) or touched-list overflow: every column from the formula
        for (int x = lane; x < cw; x += 64) { out[x] = set_jaccard_scaled(ni, a_nnz[cbase + x], cnt[x], om); cnt[x] = 0; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

}  // namespace ssg

using namespace ssg;

// A_i for rows [row0,row0+nrows): a_idx/a_val [nrows, cap] (a_val = half 1, for ssg_invert_index), a_nnz [nrows];
// rank = ssg_topk_rank(D, rowmax = half(1) everywhere, K = k) of the same rows; *overflow counts rows with more than cap members.
extern "C" int ssg_knn_sets(const uint16_t* D, const int32_t* rank, int N, int row0, int nrows, int K, int cap, int32_t* a_idx, uint16_t* a_val,
                            int32_t* a_nnz, int32_t* overflow, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || K <= 0 || K > N || cap <= 0) { ssg_set_error("ssg_knn_sets: bad shape N=%d nrows=%d K=%d cap=%d", N, nrows, K, cap); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(knn_set_kernel, dim3((nrows + 3) / 4), dim3(256), 0, stream, D, rank, N, row0, nrows, K, cap, a_idx, a_val, a_nnz, overflow);
  SSG_LAUNCH_CHECK("knn_set_kernel");
  return SSG_OK;
}

// J'[i,k] = half(half(jaccard(A_i, A_k)) * half(1-lambda)) for rows [row0,row0+nrows); a_idx/a_nnz cover ALL N rows,
// colptr/inv_row = ssg_invert_index of them.
extern "C" int ssg_set_jaccard_rows(const int32_t* a_idx, const int32_t* a_nnz, int capA, const int64_t* colptr, const int32_t* inv_row, int N, int row0,
                                    int nrows, uint16_t one_minus_lambda_half, uint16_t* Jp, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || capA <= 0) { ssg_set_error("ssg_set_jaccard_rows: bad shape"); return SSG_ERR_INVALID; }
  const int cw = N < PCHUNK ? N : PCHUNK;
  const size_t lds = (((size_t)cw * 2 + 15) & ~(size_t)15) + 16 + (size_t)PTCAP * 2 + 64;
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)set_jaccard_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int grid = nrows < 256 * per_cu ? nrows : 256 * per_cu;
  hipLaunchKernelGGL(set_jaccard_rows_kernel, dim3(grid), dim3(64), lds, stream, a_nnz, capA, a_idx, colptr, inv_row, N, row0, nrows,
                     one_minus_lambda_half, Jp);
  SSG_LAUNCH_CHECK("set_jaccard_rows_kernel");
  return SSG_OK;
}
