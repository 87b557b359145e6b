// jaccard.hip -- K8 + K9: inverted index and Jaccard distance rows.
//
// Replaces reid/rerank.py:101-122:
//   invIndex[c] = rows r with V[r,c] != 0                                   (:101-103)
//   temp_min[k] += minimum(V[i,c], V[k,c])  for c in nonzero(V[i]) ascending, half adds (:108-114)
//   jaccard[i]  = 1 - temp_min/(2 - temp_min);  clamp <0 -> 0                 (:115-118)
//   final       = jaccard*(1-lambda) + source_dist*lambda                    (:122)
// Only the compact half matrix J' = half(jaccard * half(1-lambda)) is written to HBM
// (2 bytes/entry; the f64 final_dist is rebuilt on the fly from J' and the source vector v,
// see final_dist_value()).  K9 is HBM-write-bound: 2*N^2 bytes per split.
//
// One wave per row: the accumulator row lives in LDS (half, chunked to 32768 columns);
// columns of row i are walked in ascending order (the reference's sequential half
// rounding), the entries of one inverted list are independent and spread over the lanes.
#include "ssg_common.h"

namespace ssg {

__device__ __forceinline__ void wave_sync2() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ void inv_count_kernel(const int32_t* __restrict__ q_idx, const hbits* __restrict__ q_val, const int32_t* __restrict__ q_nnz,
                                 int nrows, int capQ, int32_t* __restrict__ colcnt) {
  const int row = (int)blockIdx.x;
  if (row >= nrows) return;
  const int n = q_nnz[row];
  for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x)
    if (q_val[(int64_t)row * capQ + p] & 0x7fffu) atomicAdd(&colcnt[q_idx[(int64_t)row * capQ + p]], 1);
}

// exclusive scan of cnt[0..n) into ptr[0..n], single block; also clears cnt for reuse as cursor
__global__ __launch_bounds__(1024) void exscan_kernel(int32_t* __restrict__ cnt, int n, int64_t* __restrict__ ptr) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  for (int base = 0; base < n; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int64_t x = i < n ? (int64_t)cnt[i] : 0;
    int64_t s = x;
    for (int sh = 1; sh < 64; sh <<= 1) { const int64_t o = __shfl_up(s, sh, 64); if (lane >= sh) s += o; }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const int64_t c = carry;
    if (i < n) { ptr[i] = c + woff + s - x; cnt[i] = 0; }
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n] = carry;
}

__global__ void inv_fill_kernel(const int32_t* __restrict__ q_idx, const hbits* __restrict__ q_val, const int32_t* __restrict__ q_nnz,
                                int nrows, int capQ, const int64_t* __restrict__ colptr, int32_t* __restrict__ cursor,
                                int32_t* __restrict__ inv_row, hbits* __restrict__ inv_val) {
  const int row = (int)blockIdx.x;
  if (row >= nrows) return;
  const int n = q_nnz[row];
  for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x) {
    const hbits v = q_val[(int64_t)row * capQ + p];
    if (!(v & 0x7fffu)) continue;
    const int c = q_idx[(int64_t)row * capQ + p];
    const int64_t slot = colptr[c] + atomicAdd(&cursor[c], 1);   // order inside a column is irrelevant (distinct k)
    inv_row[slot] = row; inv_val[slot] = v;
  }
}

constexpr int JCHUNK = 32768;   // columns per LDS pass (64 KiB of half accumulators)

// J'[i,k] = half( clamp(1 - t/(2-t)) * half(1-lambda) ),  t = sequential half sum of minima
__device__ __forceinline__ hbits jaccard_scaled(hbits t, hbits om) {
  hbits j = h_sub(H_ONE, h_div(t, h_sub(H_TWO, t)));
  if (h2f(j) < 0.f) j = 0;
  return h_mul(j, om);
}

__global__ __launch_bounds__(64) void jaccard_rows_kernel(const int32_t* __restrict__ q_idx, const hbits* __restrict__ q_val,
                                                          const int32_t* __restrict__ q_nnz, int capQ,
                                                          const int64_t* __restrict__ colptr, const int32_t* __restrict__ inv_row,
                                                          const hbits* __restrict__ inv_val, int N, int row0, int nrows, hbits om,
                                                          hbits* __restrict__ Jp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  hbits* t = reinterpret_cast<hbits*>(smem);
  const int il = (int)blockIdx.x;
  if (il >= nrows) return;
  const int i = row0 + il, lane = lane_id();
  const int n = q_nnz[i];
  const int32_t* ci = q_idx + (int64_t)i * capQ;
  const hbits* cv = q_val + (int64_t)i * capQ;
  const hbits jp0 = jaccard_scaled(0, om);   // rows that share no column with i
  for (int cbase = 0; cbase < N; cbase += JCHUNK) {
    const int cw = (N - cbase) < JCHUNK ? (N - cbase) : JCHUNK;
    for (int x = lane * 8; x < cw; x += 512) *reinterpret_cast<uint4*>(t + x) = make_uint4(0, 0, 0, 0);
    wave_sync2();
    for (int p = 0; p < n; p++) {              // ascending columns of row i  (:110-114)
      const hbits vic = cv[p];
      if (!(vic & 0x7fffu)) continue;
      const int c = ci[p];
      const int64_t e0 = colptr[c], e1 = colptr[c + 1];
      for (int64_t e = e0 + lane; e < e1; e += 64) {
        const int k = inv_row[e] - cbase;
        if (k >= 0 && k < cw) {
          const hbits vkc = inv_val[e];
          const hbits mn = h2f(vkc) < h2f(vic) ? vkc : vic;
          t[k] = h_add(t[k], mn);
        }
      }
      wave_sync2();
    }
    // epilogue: accumulator -> J' row chunk
    hbits* out = Jp + (int64_t)il * N + cbase;
    const bool vec = (((int64_t)il * N + cbase) & 7) == 0;
    for (int x = lane * 8; x < cw; x += 512) {
      hbits r[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const hbits tv = (x + e < cw) ? t[x + e] : (hbits)0;
        r[e] = tv ? jaccard_scaled(tv, om) : jp0;
      }
      if (vec && x + 8 <= cw) {
        *reinterpret_cast<uint4*>(out + x) = make_uint4(r[0] | ((unsigned)r[1] << 16), r[2] | ((unsigned)r[3] << 16),
                                                        r[4] | ((unsigned)r[5] << 16), r[6] | ((unsigned)r[7] << 16));
      } else {
        for (int e = 0; e < 8; e++) if (x + e < cw) out[x + e] = r[e];
      }
    }
    wave_sync2();
  }
}

// API materialisation of rerank.py:122 (f64 N x N); not on the fused device path.
__global__ void final_dist_kernel(const hbits* __restrict__ Jp, const hbits* __restrict__ v, int N, int row0, int nrows, double lambda_value,
                                  double* __restrict__ out) {
  const int64_t total = (int64_t)nrows * N;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int il = (int)(x / N), k = (int)(x - (int64_t)il * N);
    out[x] = final_dist_value(Jp[x], v[row0 + il], v[k], lambda_value);
  }
}

}  // namespace ssg

using namespace ssg;

// inverted index of the sparse V_qe rows.  colcnt [ncols] int32 scratch (zeroed here),
// colptr [ncols+1] int64, inv_row/inv_val sized to the total nnz (<= sum q_nnz).
extern "C" int ssg_invert_index(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int nrows, int ncols, int capQ,
                                int32_t* colcnt, int64_t* colptr, int32_t* inv_row, uint16_t* inv_val, hipStream_t stream) {
  if (nrows <= 0 || ncols <= 0) { ssg_set_error("ssg_invert_index: empty"); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemsetAsync(colcnt, 0, (size_t)ncols * sizeof(int32_t), stream));
  hipLaunchKernelGGL(inv_count_kernel, dim3(nrows), dim3(64), 0, stream, q_idx, q_val, q_nnz, nrows, capQ, colcnt);
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, stream, colcnt, ncols, colptr);
  hipLaunchKernelGGL(inv_fill_kernel, dim3(nrows), dim3(64), 0, stream, q_idx, q_val, q_nnz, nrows, capQ, colptr, colcnt, inv_row, inv_val);
  SSG_LAUNCH_CHECK("invert_index");
  return SSG_OK;
}

extern "C" int ssg_jaccard_rows(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int capQ, const int64_t* colptr,
                                const int32_t* inv_row, const uint16_t* inv_val, int N, int row0, int nrows, uint16_t one_minus_lambda_half,
                                uint16_t* Jp, hipStream_t stream) {
  if (N <= 0 || nrows <= 0) { ssg_set_error("ssg_jaccard_rows: empty"); return SSG_ERR_INVALID; }
  const int cw = N < JCHUNK ? N : JCHUNK;
  const size_t lds = (((size_t)cw * 2 + 1023) / 1024) * 1024 + 16;
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)jaccard_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(jaccard_rows_kernel, dim3(nrows), dim3(64), lds, stream, q_idx, q_val, q_nnz, capQ, colptr, inv_row, inv_val, N, row0, nrows,
                     one_minus_lambda_half, Jp);
  SSG_LAUNCH_CHECK("jaccard_rows_kernel");
  return SSG_OK;
}

extern "C" int ssg_final_dist_f64(const uint16_t* Jp, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double* out,
                                  hipStream_t stream) {
  if (N <= 0 || nrows <= 0) { ssg_set_error("ssg_final_dist_f64: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(final_dist_kernel, dim3(2048), dim3(256), 0, stream, Jp, v, N, row0, nrows, lambda_value, out);
  SSG_LAUNCH_CHECK("final_dist_kernel");
  return SSG_OK;
}
