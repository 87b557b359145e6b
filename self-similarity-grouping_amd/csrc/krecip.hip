// krecip.hip -- K6 + K7: k-reciprocal encoding and local query expansion (sparse).
//
// K6 replaces reid/rerank.py:74-92: for every row i the k-reciprocal set R(i,k1), its
// 1/2-k expansion (rule at :87), np.unique, and the Gaussian weights
//   V[i, idx] = half(exp(-Dn[i,idx])) / half(sum)            (np.sum = pairwise float32)
// The reference keeps V as a dense N x N half matrix; here it is a sparse row of at most
// (k1+1)*(round(k1/2)+2) entries, sorted by column (np.unique order).
//
// K7 replaces reid/rerank.py:94-99:  V_qe[i,:] = half(float32 sum_{r<k2} V[rank[i,r],:] / k2),
// a k2-way merge of sorted sparse rows with the reference's left-to-right float32 sum.
//
// Both are latency/LDS-bound, not HBM-bound: the rank lists (N x (k1+1) int32) and V stay
// L2-resident; one wave per row, all set algebra in LDS with wave ballots (no block
// barriers, rows of different lengths never wait on each other).
#include "ssg_common.h"

namespace ssg {

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// numpy pairwise summation order (see oracle/ssg_oracle.c pairwise_sum_f32)
__device__ float pairwise_sum_f32(const float* a, int n) {
  if (n < 8) { float r = 0.f; for (int i = 0; i < n; i++) r += a[i]; return r; }
  if (n <= 128) {
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
      r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int n2 = n / 2; n2 -= n2 % 8;
  return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
}

// LDS per wave: rec[64] | expn[cap] | flag[cap] | uniq[cap] | wf[cap] (float) | wh[cap] (half)
__global__ __launch_bounds__(256) void krecip_kernel(const hbits* __restrict__ D, const unsigned* __restrict__ rowmax,
                                                     const int32_t* __restrict__ rank, int N, int row0, int nrows, int K, int K1,
                                                     int kh, int cap, int32_t* __restrict__ v_idx, hbits* __restrict__ v_val,
                                                     int32_t* __restrict__ v_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int il = (int)blockIdx.x * 4 + wave;   // row within this row block
  const int capp = (cap + 7) & ~3;              // array pitch: room for the 4-entry vector reads of step (c) past the last entry
  const size_t per_wave = 64 * 4 + (size_t)capp * 18;
  unsigned char* wbase = smem + (size_t)wave * ((per_wave + 15) & ~(size_t)15);
  int32_t* rec = reinterpret_cast<int32_t*>(wbase);
  int32_t* expn = rec + 64;
  int32_t* flag = expn + capp;
  int32_t* uniq = flag + capp;
  float* wf = reinterpret_cast<float*>(uniq + capp);
  hbits* wh = reinterpret_cast<hbits*>(wf + capp);
  if (il >= nrows) return;
  const int i = row0 + il;
  const uint64_t lt = lanemask_lt();

  // The rank lists are L2-resident, but every read of one is a ~1 us round trip and the walk below is a chain of them: the reads are
  // issued EIGHT AT A TIME (clamped index, unconditional) so that a list costs ceil(len / 8) round trips instead of len (round 4: the
  // one-load-per-iteration loops made this kernel 0.23 ms of pure latency at N = 16 000).
  auto holds = [&](const int32_t* __restrict__ list, int len, int what) -> bool {     // is `what` among list[0 .. len)?
    bool h = false;
    for (int b0 = 0; b0 < len; b0 += 8) {
      int x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = list[min(b0 + u, len - 1)];
#pragma unroll
      for (int u = 0; u < 8; u++) h |= (x[u] == what);
    }
    return h;
  };
  // the same test on 16-byte pieces of a list (rank rows are 4-byte aligned only: a vector type with that alignment): a lane's reads of ONE
  // list then touch one or two cache lines per instruction instead of one line per 4-byte element -- with every lane on a different
  // list the texture addresser serves a wave's load line by line, and the eleven element loads per (candidate, neighbour) pair made
  // step (b) 0.14 of the kernel's 0.19 ms (ablation builds, round 4)
  typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
  auto holds4 = [&](const int32_t* __restrict__ list, int len, int what) -> bool {     // needs list[0 .. roundup4(len)) readable
    bool h = false;
    for (int b0 = 0; b0 < len; b0 += 16) {
      v4i_a4 x[4];
#pragma unroll
      for (int u = 0; u < 4; u++) x[u] = *reinterpret_cast<const v4i_a4*>(list + min(b0 + 4 * u, ((len + 3) & ~3) - 4));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int b = min(b0 + 4 * u, ((len + 3) & ~3) - 4);
        h |= (b < len && x[u].x == what) | (b + 1 < len && x[u].y == what) | (b + 2 < len && x[u].z == what) | (b + 3 < len && x[u].w == what);
      }
    }
    return h;
  };
  // (a) k-reciprocal neighbours  (rerank.py:76-79)
  const int f = lane < K1 ? rank[(int64_t)i * K + lane] : -1;
  const bool hit = lane < K1 && holds(rank + (int64_t)f * K, K1, i);
  const uint64_t rmask = __ballot(hit);
  const int nrec = __popcll(rmask);
  if (hit) { const int p = __popcll(rmask & lt); rec[p] = f; expn[p] = f; }
  int ne = nrec;
  wave_sync();

  // (b) 1/2-k expansion  (rerank.py:81-88).  All (candidate, neighbour) pairs at once: first every candidate's first kh neighbours
  // (one round trip), then for every such neighbour whether the candidate is among ITS first kh (two round trips for kh <= 16) -- the
  // per-candidate decisions that follow only touch LDS.  `uniq` / `flag` are free until step (c): cfs = uniq, chf = flag.
  int32_t* cfs = uniq;
  int32_t* chf = flag;
  const int npair = nrec * kh;
  for (int idx = lane; idx < npair; idx += 64) {
    const int a = idx / kh, l = idx - a * kh;
    cfs[idx] = rank[(int64_t)rec[a] * K + l];
  }
  wave_sync();
  for (int idx = lane; idx < npair; idx += 64) {
    const int a = idx / kh;
    const int32_t* nl = rank + (int64_t)cfs[idx] * K;
    chf[idx] = (((kh + 3) & ~3) <= K ? holds4(nl, kh, rec[a]) : holds(nl, kh, rec[a])) ? 1 : 0;
  }
  wave_sync();
  const int recl = lane < nrec ? rec[lane] : -1;        // lane q holds rec[q] (nrec <= K1 <= 64): membership tests read it with v_readlane
  for (int a = 0; a < nrec; a++) {
    const int cf = lane < kh ? cfs[a * kh + lane] : -1;
    const bool chit = lane < kh && chf[a * kh + lane] != 0;
    const uint64_t cmask = __ballot(chit);
    const int nc = __popcll(cmask);
    bool inrec = false;
    for (int q = 0; q < nrec; q++) inrec |= (__builtin_amdgcn_readlane(recl, q) == cf);
    inrec &= chit;
    const int inter = __popcll(__ballot(inrec));
    if ((double)inter > (2.0 / 3.0) * (double)nc) {   // len(intersect1d) > 2/3*len(candidate set)
      if (chit) expn[ne + __popcll(cmask & lt)] = cf;
      ne += nc;
    }
  }
  wave_sync();

  // (c) np.unique: sorted distinct columns  (rerank.py:90).  Counting sort by comparison, on 4-entry LDS vectors: an entry is kept when
  // no EARLIER entry equals it; a kept entry's position is the number of kept entries smaller than it.  (Round 4: the one-entry-per-trip
  // loops of the first version were ~1500 dependent LDS reads per lane -- 100 us per row, the whole kernel's time.)
  const int ne4 = (ne + 3) & ~3;
  if (lane < ne4 - ne) expn[ne + lane] = 0x7fffffff;       // sentinels up to the vector boundary (never equal to / smaller than a column)
  wave_sync();
  for (int p0 = 0; p0 < ne; p0 += 64) {
    const int p = p0 + lane;
    const int x = p < ne ? expn[p] : 0x7fffffff;
    bool dup = false;
    const int qend = min(p0 + 64, ne4);                      // entries at or after p never count: masked below
    for (int q0 = 0; q0 < qend; q0 += 16) {                  // four vectors in flight (reads past qend stay inside the padded array: masked by q < p)
      int4 e[4];
#pragma unroll
      for (int u = 0; u < 4; u++) e[u] = *reinterpret_cast<const int4*>(expn + min(q0 + 4 * u, ne4 - 4));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = min(q0 + 4 * u, ne4 - 4);
        dup |= (q < p && e[u].x == x) | (q + 1 < p && e[u].y == x) | (q + 2 < p && e[u].z == x) | (q + 3 < p && e[u].w == x);
      }
    }
    if (p < ne4) flag[p] = (p < ne && !dup) ? x : 0x7fffffff;   // kept entries keep their column, duplicates and padding become +inf
  }
  wave_sync();
  int nu = 0;
  for (int p0 = 0; p0 < ne; p0 += 64) {
    const int p = p0 + lane;
    const int x = p < ne ? flag[p] : 0x7fffffff;
    const bool kept = x != 0x7fffffff;
    int pos = 0;
    for (int q0 = 0; q0 < ne4; q0 += 16) {
      int4 e[4];
#pragma unroll
      for (int u = 0; u < 4; u++) e[u] = *reinterpret_cast<const int4*>(flag + min(q0 + 4 * u, ne4 - 4));
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (q0 + 4 * u < ne4) pos += (e[u].x < x) + (e[u].y < x) + (e[u].z < x) + (e[u].w < x);
    }
    if (kept) uniq[pos] = x;
    nu += __popcll(__ballot(kept));
  }
  wave_sync();

  // (d) weights  (rerank.py:91-92); Dn[i,idx] = half(D[i,idx] / rowmax[i]) recomputed on the fly
  const float fmx = h2f((hbits)rowmax[il]);
  const hbits* drow = D + (int64_t)il * N;
  for (int p = lane; p < nu; p += 64) {
    const hbits dn = f2h(h2f(drow[uniq[p]]) / fmx);
    const hbits w = h_exp_neg(dn);
    wh[p] = w; wf[p] = h2f(w);
  }
  wave_sync();
  float s = 0.f;
  if (lane == 0) s = pairwise_sum_f32(wf, nu);
  s = __shfl(s, 0, 64);
  const hbits sum16 = f2h(s);
  for (int p = lane; p < nu; p += 64) {
    v_idx[(int64_t)il * cap + p] = uniq[p];
    v_val[(int64_t)il * cap + p] = h_div(wh[p], sum16);
  }
  if (lane == 0) v_nnz[il] = nu;
}

// ---------------------------------------------------------------------------------- K7
// lower_bound on a sorted int list in LDS
__device__ __forceinline__ int lower_bound_i32(const int32_t* a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
  return lo;
}

// LDS per wave: for each of k2 lists: idx[capV] int32 | val[capV] half | hpre[capV+1] int32
__global__ __launch_bounds__(256) void query_expand_kernel(const int32_t* __restrict__ v_idx, const hbits* __restrict__ v_val,
                                                           const int32_t* __restrict__ v_nnz, const int32_t* __restrict__ rank,
                                                           int row0, int nrows, int K, int kk, int capV, int capQ, int capL,
                                                           int32_t* __restrict__ q_idx, hbits* __restrict__ q_val,
                                                           int32_t* __restrict__ q_nnz, int32_t* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int il = (int)blockIdx.x * 4 + wave;
  // capL = longest source row actually present (LDS is sized for it, not for the worst-case capV)
  const size_t per_list = (size_t)capL * 4 + (size_t)(capL + 1) * 4 + (((size_t)capL * 2 + 3) & ~(size_t)3);
  const size_t per_wave = (per_list * kk + 64 + 15) & ~(size_t)15;
  unsigned char* wbase = smem + (size_t)wave * per_wave;
  if (il >= nrows) return;
  const int i = row0 + il;
  auto L_idx = [&](int r) { return reinterpret_cast<int32_t*>(wbase + per_list * r); };
  auto L_pre = [&](int r) { return reinterpret_cast<int32_t*>(wbase + per_list * r) + capL; };
  auto L_val = [&](int r) { return reinterpret_cast<hbits*>(wbase + per_list * r + (size_t)capL * 4 + (size_t)(capL + 1) * 4); };
  int32_t* nn = reinterpret_cast<int32_t*>(wbase + per_list * kk);   // list lengths

  // stage the kk source rows (V rows of the first k2 ranked neighbours, rerank.py:97): sources and lengths of all lists first (two
  // round trips for the whole row instead of two per list), then the entries
  int nmax = 0;
  {
    const int srcl = lane < kk ? rank[(int64_t)i * K + lane] : 0;
    int nl = lane < kk ? v_nnz[srcl] : 0;
    int m = nl;
    for (int sh = 1; sh < 64; sh <<= 1) m = max(m, __shfl_xor(m, sh, 64));
    nmax = m;
    if (nl > capL) {                     // a guessed max_nnz was too small: flag it (the caller redoes the step), keep the row in bounds
      if (overflow) atomicMax(overflow, nl);
      nl = capL;
    }
    if (lane < kk) nn[lane] = nl;
    // the entries of ALL lists as one flat index space, four per lane in flight (list by list, every list paid its own round trip)
    int incl = nl;
    for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_up(incl, sh, 64); if (lane >= sh) incl += o; }
    const int total = __shfl(incl, kk - 1, 64);
    for (int f0 = 0; f0 < total; f0 += 256) {
      int rr[4], pp[4], ci[4]; hbits cv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = f0 + u * 64 + lane;
        // list holding flat entry f: the last list whose start is <= f (list lengths / sources are wave-uniform per list: v_readlane,
        // not the LDS crossbar)
        int r = 0, start = 0, src = __builtin_amdgcn_readlane(srcl, 0);
        for (int q = 1; q < kk; q++) {
          const int sq = __builtin_amdgcn_readlane(incl, q - 1);
          if (f >= sq) { r = q; start = sq; src = __builtin_amdgcn_readlane(srcl, q); }
        }
        rr[u] = r; pp[u] = f - start;
        const int64_t a = (int64_t)src * capV + (f < total ? pp[u] : 0);
        ci[u] = v_idx[a]; cv[u] = v_val[a];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (f0 + u * 64 + lane < total) { L_idx(rr[u])[pp[u]] = ci[u]; L_val(rr[u])[pp[u]] = cv[u]; }
    }
  }
  // overflow[1] = longest V row met so far (feeds the caller's next guess); a plain read first: after the first waves almost no atomics
  if (lane == 0 && overflow && nmax > *reinterpret_cast<volatile int32_t*>(overflow + 1)) atomicMax(overflow + 1, nmax);
  wave_sync();
  // head flags: element (r,p) is the head of its column iff no earlier list holds the column;
  // hpre[r][p] = number of heads among list r's first p entries
  for (int r = 0; r < kk; r++) {
    const int n = nn[r];
    int run = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
      const int p = p0 + lane;
      bool head = false;
      if (p < n) {
        const int c = L_idx(r)[p];
        head = true;
        for (int r2 = 0; r2 < r; r2++) {
          const int n2 = nn[r2];
          const int q = lower_bound_i32(L_idx(r2), n2, c);
          if (q < n2 && L_idx(r2)[q] == c) head = false;
        }
      }
      const uint64_t hm = __ballot(head);
      if (p < n) L_pre(r)[p] = run + __popcll(hm & lanemask_lt());
      // remember head flag in the sign of nothing: recomputed below via hpre differences
      run += __popcll(hm);
    }
    if (lane == 0) L_pre(r)[n] = run;
  }
  wave_sync();
  int total = 0;
  for (int r = 0; r < kk; r++) total += L_pre(r)[nn[r]];
  // emit heads: output slot = number of head columns smaller than c; value = sequential
  // float32 sum over r = 0..k2-1 (np.mean(axis=0) order), / k2, -> half
  const float fk = (float)kk;
  for (int r = 0; r < kk; r++) {
    const int n = nn[r];
    for (int p = lane; p < n; p += 64) {
      const bool head = (L_pre(r)[p + 1] - L_pre(r)[p]) != 0;
      if (!head) continue;
      const int c = L_idx(r)[p];
      int pos = 0;
      float s = 0.f;
      bool started = false;
      for (int r2 = 0; r2 < kk; r2++) {
        const int n2 = nn[r2];
        const int q = lower_bound_i32(L_idx(r2), n2, c);
        pos += L_pre(r2)[q];
        const float x = (q < n2 && L_idx(r2)[q] == c) ? h2f(L_val(r2)[q]) : 0.f;
        if (!started) { s = x; started = true; } else s += x;
      }
      q_idx[(int64_t)il * capQ + pos] = c;
      q_val[(int64_t)il * capQ + pos] = f2h(s / fk);
    }
  }
  if (lane == 0) q_nnz[il] = total;
}

}  // namespace ssg

using namespace ssg;

static inline int round_half_even_div2(int k1) { return (k1 % 2 == 0) ? k1 / 2 : ((k1 / 2) % 2 == 0 ? k1 / 2 : k1 / 2 + 1); }

// capacity of one sparse V row for a given k1 (entries): (k1+1) * (round(k1/2) + 2)
extern "C" int ssg_krecip_row_capacity(int k1) { return (k1 + 1) * (round_half_even_div2(k1) + 2); }

extern "C" int ssg_krecip(const uint16_t* D, const uint32_t* rowmax, const int32_t* rank, int N, int row0, int nrows, int K, int k1,
                          int cap, int32_t* v_idx, uint16_t* v_val, int32_t* v_nnz, hipStream_t stream) {
  int K1 = k1 + 1; if (K1 > N) K1 = N; if (K1 > K) K1 = K;
  int kh = round_half_even_div2(k1) + 1; if (kh > N) kh = N; if (kh > K) kh = K;
  if (K1 <= 0 || K1 > 64 || cap < K1 + K1 * kh) {
    ssg_set_error("ssg_krecip: need k1+1 <= 64 and cap >= %d (got %d)", K1 + K1 * kh, cap);
    return SSG_ERR_INVALID;
  }
  const size_t per_wave = ((64 * 4 + (size_t)((cap + 7) & ~3) * 18) + 15) & ~(size_t)15;
  const size_t lds = per_wave * 4;
  if (lds > 160 * 1024) { ssg_set_error("ssg_krecip: k1=%d needs %zu B LDS", k1, lds); return SSG_ERR_INVALID; }
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)krecip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(krecip_kernel, dim3((nrows + 3) / 4), dim3(256), lds, stream, D, rowmax, rank, N, row0, nrows, K, K1, kh, cap, v_idx,
                     v_val, v_nnz);
  SSG_LAUNCH_CHECK("krecip_kernel");
  return SSG_OK;
}

extern "C" int ssg_query_expand(const int32_t* v_idx, const uint16_t* v_val, const int32_t* v_nnz, const int32_t* rank, int N, int row0,
                                int nrows, int K, int k2, int capV, int capQ, int max_nnz, int32_t* q_idx, uint16_t* q_val, int32_t* q_nnz,
                                int32_t* overflow, hipStream_t stream) {
  int kk = k2; if (kk > N) kk = N; if (kk > K) kk = K;
  const int capL = max_nnz < 1 ? 1 : max_nnz;
  if (kk <= 0 || capL > capV || capQ < kk * capL) {
    ssg_set_error("ssg_query_expand: need max_nnz (%d) <= capV (%d) and capQ (%d) >= k2*max_nnz (%d)", max_nnz, capV, capQ, kk * capL);
    return SSG_ERR_INVALID;
  }
  const size_t per_list = (size_t)capL * 4 + (size_t)(capL + 1) * 4 + (((size_t)capL * 2 + 3) & ~(size_t)3);
  const size_t per_wave = (per_list * kk + 64 + 15) & ~(size_t)15;
  const size_t lds = per_wave * 4;
  if (lds > 160 * 1024) { ssg_set_error("ssg_query_expand: k2=%d max_nnz=%d needs %zu B LDS", k2, capL, lds); return SSG_ERR_INVALID; }
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)query_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(query_expand_kernel, dim3((nrows + 3) / 4), dim3(256), lds, stream, v_idx, v_val, v_nnz, rank, row0, nrows, K, kk, capV,
                     capQ, capL, q_idx, q_val, q_nnz, overflow);
  SSG_LAUNCH_CHECK("query_expand_kernel");
  return SSG_OK;
}
