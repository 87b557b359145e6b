// rerank_init.hip -- float32 k-reciprocal re-ranking ("re_ranking_init").
//
// Replaces reid/rerank.py:171-234 == reid/rerank_initial.py:40-99 (caller: reid/eug.py:223-226):
// the cosine form original_dist = 2 - 2 x.y over the stacked [query; gallery] features, row
// normalisation, top-(k1+1) ranking (np.argpartition(.., range(1,k1+1)) = the k1+1 smallest in
// ascending order), k-reciprocal encoding with float32 weights, k2 query expansion, Jaccard
// distance for the query rows and the blend  final = J*(1-lambda) + original_dist*lambda,
// returned as the [query, gallery] block.  Everything is float32 like the reference (numpy
// promotes nothing here); np.dot / np.exp are not reproducible bit for bit, so parity for this
// variant is tolerance based (tests: 2e-5).
//
// The Gram matrix comes from the fp32-MFMA GEMM of conv.hip (epilogue 2 - 2*acc).  The sparse
// stages mirror krecip.hip / jaccard.hip with float values; this path serves the evaluation-size
// problems of the semi-supervised driver and is written for clarity, not tuned like the half path.
#include "ssg_common.h"

namespace ssg {

__device__ __forceinline__ void wsync_i() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// monotone map float -> uint32 (handles the slightly negative self distances 2 - 2|x|^2)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void affine_2m2x_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 2.f - 2.f * in[i];
}

// rowmax[i] = max_j D[i,j]  (== np.max(original_dist, axis=0) for the symmetric D)
__global__ __launch_bounds__(256) void rowmax_f32_kernel(const float* __restrict__ D, int N, float* __restrict__ rowmax) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= N) return;
  const int lane = lane_id();
  float m = -INFINITY;
  for (int j = lane; j < N; j += 64) m = fmaxf(m, D[(int64_t)row * N + j]);
  for (int sh = 1; sh < 64; sh <<= 1) m = fmaxf(m, __shfl_xor(m, sh, 64));
  if (lane == 0) rowmax[row] = m;
}

// rank[i, 0:K] = columns of the K smallest D[i,:]/rowmax[i] in (value, column) order
__global__ __launch_bounds__(256) void topk_f32_kernel(const float* __restrict__ D, const float* __restrict__ rowmax, int N, int K,
                                                       int32_t* __restrict__ rank) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= N) return;
  const int lane = lane_id();
  const float mx = rowmax[row];
  uint64_t mine = ~0ULL, tau = ~0ULL;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    uint64_t comp = ~0ULL;
    if (j < N) comp = ((uint64_t)fkey(D[(int64_t)row * N + j] / mx) << 32) | (uint32_t)j;
    bool cand = comp < tau;
    uint64_t mask = __ballot(cand);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      const uint64_t cc = __shfl(comp, src, 64);
      const bool lt = mine < cc;
      const uint64_t up = __shfl_up(mine, 1, 64);
      const int pos = __popcll(__ballot(lt));
      mine = lt ? mine : (lane == pos ? cc : up);
      tau = __shfl(mine, K - 1, 64);
      if (lane == src) cand = false;
      cand = cand && (comp < tau);
      mask = __ballot(cand);
    }
  }
  if (lane < K) rank[(int64_t)row * K + lane] = (int32_t)(mine & 0xffffffffULL);
}

__device__ float pairwise_sum_f32_i(const float* a, int n) {   // numpy pairwise summation
  if (n < 8) { float r = 0.f; for (int i = 0; i < n; i++) r += a[i]; return r; }
  if (n <= 128) {
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) { r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7]; }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int n2 = n / 2; n2 -= n2 % 8;
  return pairwise_sum_f32_i(a, n2) + pairwise_sum_f32_i(a + n2, n - n2);
}

// k-reciprocal sets + 1/2-k expansion + weights (rerank.py:192-204), float32 values.
// LDS per wave: rec[64] | expn[cap] | flag[cap] | uniq[cap] | wf[cap]
__global__ __launch_bounds__(256) void krecip_f32_kernel(const float* __restrict__ D, const float* __restrict__ rowmax, const int32_t* __restrict__ rank,
                                                         int N, int K, int K1, int kh, int cap, int32_t* __restrict__ v_idx, float* __restrict__ v_val,
                                                         int32_t* __restrict__ v_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = (int)blockIdx.x * 4 + wave;
  const size_t per_wave = ((64 * 4 + (size_t)cap * 16) + 15) & ~(size_t)15;
  unsigned char* wbase = smem + (size_t)wave * per_wave;
  int32_t* rec = reinterpret_cast<int32_t*>(wbase);
  int32_t* expn = rec + 64; int32_t* flag = expn + cap; int32_t* uniq = flag + cap;
  float* wf = reinterpret_cast<float*>(uniq + cap);
  if (i >= N) return;
  const uint64_t lt = lanemask_lt();
  const int f = lane < K1 ? rank[(int64_t)i * K + lane] : -1;
  bool hit = false;
  if (lane < K1) { const int32_t* bw = rank + (int64_t)f * K; for (int b = 0; b < K1; b++) hit |= (bw[b] == i); }
  const uint64_t rmask = __ballot(hit);
  const int nrec = __popcll(rmask);
  if (hit) { const int p = __popcll(rmask & lt); rec[p] = f; expn[p] = f; }
  int ne = nrec;
  wsync_i();
  for (int a = 0; a < nrec; a++) {
    const int cand = rec[a];
    const int cf = lane < kh ? rank[(int64_t)cand * K + lane] : -1;
    bool chit = false;
    if (lane < kh) { const int32_t* cb = rank + (int64_t)cf * K; for (int c = 0; c < kh; c++) chit |= (cb[c] == cand); }
    const uint64_t cmask = __ballot(chit);
    const int nc = __popcll(cmask);
    bool inrec = false;
    if (chit) for (int q = 0; q < nrec; q++) inrec |= (rec[q] == cf);
    const int inter = __popcll(__ballot(inrec));
    if ((double)inter > (2.0 / 3.0) * (double)nc) { if (chit) expn[ne + __popcll(cmask & lt)] = cf; ne += nc; }
  }
  wsync_i();
  for (int p = lane; p < ne; p += 64) { const int x = expn[p]; bool first = true; for (int q = 0; q < p; q++) first &= (expn[q] != x); flag[p] = first ? 1 : 0; }
  wsync_i();
  int nu = 0;
  for (int p0 = 0; p0 < ne; p0 += 64) {
    const int p = p0 + lane; bool isf = false;
    if (p < ne && flag[p]) { isf = true; const int x = expn[p]; int pos = 0; for (int q = 0; q < ne; q++) pos += (flag[q] && expn[q] < x); uniq[pos] = x; }
    nu += __popcll(__ballot(isf));
  }
  wsync_i();
  const float mx = rowmax[i];
  for (int p = lane; p < nu; p += 64) wf[p] = expf(-(D[(int64_t)i * N + uniq[p]] / mx));
  wsync_i();
  float s = 0.f;
  if (lane == 0) s = pairwise_sum_f32_i(wf, nu);
  s = __shfl(s, 0, 64);
  for (int p = lane; p < nu; p += 64) { v_idx[(int64_t)i * cap + p] = uniq[p]; v_val[(int64_t)i * cap + p] = wf[p] / s; }
  if (lane == 0) v_nnz[i] = nu;
}

__device__ __forceinline__ int lb_i32(const int32_t* a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
  return lo;
}

// V_qe[i,:] = mean_{r<k2} V[rank[i,r],:]  (rerank.py:207-212), float32; one wave per row.
// LDS per wave: per list idx[capL] | pre[capL+1] | val[capL]
__global__ __launch_bounds__(256) void query_expand_f32_kernel(const int32_t* __restrict__ v_idx, const float* __restrict__ v_val, const int32_t* __restrict__ v_nnz,
                                                               const int32_t* __restrict__ rank, int N, int K, int kk, int capV, int capQ, int capL,
                                                               int32_t* __restrict__ q_idx, float* __restrict__ q_val, int32_t* __restrict__ q_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = (int)blockIdx.x * 4 + wave;
  const size_t per_list = (size_t)capL * 4 + (size_t)(capL + 1) * 4 + (size_t)capL * 4;
  const size_t per_wave = (per_list * kk + 64 + 15) & ~(size_t)15;
  unsigned char* wbase = smem + (size_t)wave * per_wave;
  if (i >= N) return;
  auto L_idx = [&](int r) { return reinterpret_cast<int32_t*>(wbase + per_list * r); };
  auto L_pre = [&](int r) { return reinterpret_cast<int32_t*>(wbase + per_list * r) + capL; };
  auto L_val = [&](int r) { return reinterpret_cast<float*>(wbase + per_list * r + (size_t)capL * 4 + (size_t)(capL + 1) * 4); };
  int32_t* nn = reinterpret_cast<int32_t*>(wbase + per_list * kk);
  for (int r = 0; r < kk; r++) {
    const int src = rank[(int64_t)i * K + r];
    const int n = v_nnz[src];
    if (lane == 0) nn[r] = n;
    for (int p = lane; p < n; p += 64) { L_idx(r)[p] = v_idx[(int64_t)src * capV + p]; L_val(r)[p] = v_val[(int64_t)src * capV + p]; }
  }
  wsync_i();
  for (int r = 0; r < kk; r++) {
    const int n = nn[r]; int run = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
      const int p = p0 + lane; bool head = false;
      if (p < n) {
        const int c = L_idx(r)[p]; head = true;
        for (int r2 = 0; r2 < r; r2++) { const int n2 = nn[r2]; const int q = lb_i32(L_idx(r2), n2, c); if (q < n2 && L_idx(r2)[q] == c) head = false; }
      }
      const uint64_t hm = __ballot(head);
      if (p < n) L_pre(r)[p] = run + __popcll(hm & lanemask_lt());
      run += __popcll(hm);
    }
    if (lane == 0) L_pre(r)[n] = run;
  }
  wsync_i();
  int total = 0;
  for (int r = 0; r < kk; r++) total += L_pre(r)[nn[r]];
  const float fk = (float)kk;
  for (int r = 0; r < kk; r++) {
    const int n = nn[r];
    for (int p = lane; p < n; p += 64) {
      if ((L_pre(r)[p + 1] - L_pre(r)[p]) == 0) continue;
      const int c = L_idx(r)[p];
      int pos = 0; float s = 0.f; bool started = false;
      for (int r2 = 0; r2 < kk; r2++) {
        const int n2 = nn[r2]; const int q = lb_i32(L_idx(r2), n2, c);
        pos += L_pre(r2)[q];
        const float x = (q < n2 && L_idx(r2)[q] == c) ? L_val(r2)[q] : 0.f;
        if (!started) { s = x; started = true; } else s += x;
      }
      q_idx[(int64_t)i * capQ + pos] = c; q_val[(int64_t)i * capQ + pos] = s / fk;
    }
  }
  if (lane == 0) q_nnz[i] = total;
}

__global__ void inv_count_f32_kernel(const int32_t* __restrict__ q_idx, const float* __restrict__ q_val, const int32_t* __restrict__ q_nnz, int nrows, int capQ,
                                     int32_t* __restrict__ colcnt) {
  const int row = (int)blockIdx.x; if (row >= nrows) return;
  const int n = q_nnz[row];
  for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x) if (q_val[(int64_t)row * capQ + p] != 0.f) atomicAdd(&colcnt[q_idx[(int64_t)row * capQ + p]], 1);
}
__global__ void inv_fill_f32_kernel(const int32_t* __restrict__ q_idx, const float* __restrict__ q_val, const int32_t* __restrict__ q_nnz, int nrows, int capQ,
                                    const int64_t* __restrict__ colptr, int32_t* __restrict__ cursor, int32_t* __restrict__ inv_row, float* __restrict__ inv_val) {
  const int row = (int)blockIdx.x; if (row >= nrows) return;
  const int n = q_nnz[row];
  for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x) {
    const float v = q_val[(int64_t)row * capQ + p]; if (v == 0.f) continue;
    const int c = q_idx[(int64_t)row * capQ + p];
    const int64_t slot = colptr[c] + atomicAdd(&cursor[c], 1);
    inv_row[slot] = row; inv_val[slot] = v;
  }
}

constexpr int ICHUNK = 8192;   // float accumulator columns per LDS pass (32 KiB)

// Jaccard + blend for query row i (rerank.py:218-233): out[i, g] for gallery columns g.
__global__ __launch_bounds__(64) void jaccard_init_kernel(const int32_t* __restrict__ q_idx, const float* __restrict__ q_val, const int32_t* __restrict__ q_nnz, int capQ,
                                                          const int64_t* __restrict__ colptr, const int32_t* __restrict__ inv_row, const float* __restrict__ inv_val,
                                                          const float* __restrict__ D, const float* __restrict__ rowmax, int N, int nq, float lambda_value,
                                                          float* __restrict__ out) {
  __shared__ float t[ICHUNK];
  const int i = (int)blockIdx.x; if (i >= nq) return;
  const int lane = lane_id();
  const int n = q_nnz[i], ng = N - nq;
  const float mx = rowmax[i], om = 1.f - lambda_value;
  for (int cbase = 0; cbase < N; cbase += ICHUNK) {
    const int cw = (N - cbase) < ICHUNK ? (N - cbase) : ICHUNK;
    if (cbase + cw <= nq) continue;                      // only gallery columns are returned
    for (int x = lane; x < cw; x += 64) t[x] = 0.f;
    wsync_i();
    for (int p = 0; p < n; p++) {
      const float vic = q_val[(int64_t)i * capQ + p];
      if (vic == 0.f) continue;
      const int c = q_idx[(int64_t)i * capQ + p];
      for (int64_t e = colptr[c] + lane; e < colptr[c + 1]; e += 64) {
        const int kk = inv_row[e] - cbase;
        if (kk >= 0 && kk < cw) t[kk] = t[kk] + fminf(vic, inv_val[e]);
      }
      wsync_i();
    }
    for (int x = lane; x < cw; x += 64) {
      const int k = cbase + x;
      if (k < nq) continue;
      const float j = 1.f - t[x] / (2.f - t[x]);
      out[(int64_t)i * ng + (k - nq)] = j * om + (D[(int64_t)i * N + k] / mx) * lambda_value;
    }
    wsync_i();
  }
}

}  // namespace ssg

using namespace ssg;

extern "C" int ssg_affine_2m2x_f32(const float* in, float* out, int64_t n, hipStream_t stream) {
  if (n <= 0) { ssg_set_error("ssg_affine_2m2x_f32: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(affine_2m2x_kernel, dim3(2048), dim3(256), 0, stream, in, out, n);
  SSG_LAUNCH_CHECK("affine_2m2x_kernel");
  return SSG_OK;
}

// Sparse stages of re_ranking_init on a precomputed float32 distance matrix D [N,N] (symmetric).
// Workspace (caller): rowmax [N] f32, rank [N,K] i32, v_idx/v_val [N,capV], v_nnz [N], q_idx/q_val
// [N,capQ], q_nnz [N], colcnt [N] i32, colptr [N+1] i64, inv_row/inv_val [inv_cap].  Two calls:
// stage 1 (ranking + V + V_qe) -> host reads max nnz / total nnz to size the rest; stage 2 = index + Jaccard.
extern "C" int ssg_rerank_init_stage1(const float* D, int N, int k1, int k2, int capV, float* rowmax, int32_t* rank, int32_t* v_idx, float* v_val,
                                      int32_t* v_nnz, hipStream_t stream) {
  const int K = (k1 + 1 < N) ? k1 + 1 : N;
  const int khr = (k1 % 2 == 0) ? k1 / 2 : ((k1 / 2) % 2 == 0 ? k1 / 2 : k1 / 2 + 1);
  int kh = khr + 1; if (kh > K) kh = K;
  if (N < 2 || K > 64 || capV < K + K * kh) { ssg_set_error("ssg_rerank_init_stage1: need k1+1 <= 64, capV >= %d", K + K * kh); return SSG_ERR_INVALID; }
  (void)k2;
  hipLaunchKernelGGL(rowmax_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, D, N, rowmax);
  hipLaunchKernelGGL(topk_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, D, rowmax, N, K, rank);
  const size_t per_wave = ((64 * 4 + (size_t)capV * 16) + 15) & ~(size_t)15;
  const size_t lds = per_wave * 4;
  if (lds > 160 * 1024) { ssg_set_error("ssg_rerank_init_stage1: k1 too large for LDS"); return SSG_ERR_INVALID; }
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)krecip_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(krecip_f32_kernel, dim3((N + 3) / 4), dim3(256), lds, stream, D, rowmax, rank, N, K, K, kh, capV, v_idx, v_val, v_nnz);
  SSG_LAUNCH_CHECK("rerank_init_stage1");
  return SSG_OK;
}

extern "C" int ssg_rerank_init_expand(const int32_t* v_idx, const float* v_val, const int32_t* v_nnz, const int32_t* rank, int N, int k1, int k2, int capV,
                                      int capQ, int max_nnz, int32_t* q_idx, float* q_val, int32_t* q_nnz, hipStream_t stream) {
  const int K = (k1 + 1 < N) ? k1 + 1 : N;
  int kk = k2; if (kk > N) kk = N; if (kk > K) kk = K;
  const int capL = max_nnz < 1 ? 1 : max_nnz;
  if (kk <= 0 || capL > capV || capQ < kk * capL) { ssg_set_error("ssg_rerank_init_expand: bad capacities"); return SSG_ERR_INVALID; }
  const size_t per_list = (size_t)capL * 4 + (size_t)(capL + 1) * 4 + (size_t)capL * 4;
  const size_t lds = ((per_list * kk + 64 + 15) & ~(size_t)15) * 4;
  if (lds > 160 * 1024) { ssg_set_error("ssg_rerank_init_expand: needs %zu B LDS", lds); return SSG_ERR_INVALID; }
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)query_expand_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(query_expand_f32_kernel, dim3((N + 3) / 4), dim3(256), lds, stream, v_idx, v_val, v_nnz, rank, N, K, kk, capV, capQ, capL, q_idx, q_val, q_nnz);
  SSG_LAUNCH_CHECK("query_expand_f32_kernel");
  return SSG_OK;
}

extern "C" int ssg_rerank_init_jaccard(const float* D, const float* rowmax, const int32_t* q_idx, const float* q_val, const int32_t* q_nnz, int capQ, int N,
                                       int nq, float lambda_value, int32_t* colcnt, int64_t* colptr, int32_t* inv_row, float* inv_val, float* out,
                                       hipStream_t stream) {
  if (N < 2 || nq <= 0 || nq >= N) { ssg_set_error("ssg_rerank_init_jaccard: need 0 < nq < N"); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemsetAsync(colcnt, 0, (size_t)N * sizeof(int32_t), stream));
  hipLaunchKernelGGL(inv_count_f32_kernel, dim3(N), dim3(64), 0, stream, q_idx, q_val, q_nnz, N, capQ, colcnt);
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, stream, colcnt, N, colptr);
  hipLaunchKernelGGL(inv_fill_f32_kernel, dim3(N), dim3(64), 0, stream, q_idx, q_val, q_nnz, N, capQ, colptr, colcnt, inv_row, inv_val);
  hipLaunchKernelGGL(jaccard_init_kernel, dim3(nq), dim3(64), 0, stream, q_idx, q_val, q_nnz, capQ, colptr, inv_row, inv_val, D, rowmax, N, nq, lambda_value, out);
  SSG_LAUNCH_CHECK("rerank_init_jaccard");
  return SSG_OK;
}
