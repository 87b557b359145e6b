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
#include <cstdlib>

#ifndef SSG_JAC_NT_STORE
#define SSG_JAC_NT_STORE 1         // the streaming pass's stores of J' (every line once, 2 N^2 bytes) with the nt cache policy: they need no line of the L2 the
                                   // inverted lists / V rows of the walk go through (0.613 -> 0.579 ms beside the source term at N = 16 000, 1.54 -> 1.37 at 30 000)
#endif
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

// exclusive scan of cnt[0..n) into ptr[0..n], single block; also clears cnt for reuse as cursor.
// Every wave owns one contiguous stretch (a multiple of 64 entries): pass 1 sums it (independent coalesced loads), the 16 totals become
// wave offsets, pass 2 reads the stretch again (L2) and scans it 64 entries at a time with the running carry in a register.  (Until
// round 5 the whole block walked the array 1024 entries at a time, each step a load + three barriers behind the previous one's carry:
// 25 us for 16 000 entries, twice per grouping leg.)
__global__ __launch_bounds__(1024) void exscan_kernel(int32_t* __restrict__ cnt, int n, int64_t* __restrict__ ptr) {
  __shared__ int64_t wsum[16];
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int per = (((n + 15) / 16) + 63) & ~63;                  // entries per wave
  const int lo = wave * per, hi = lo + per < n ? lo + per : n;
  if (per <= 1024) {
    // up to 16 384 entries: a wave's stretch (16 pieces of 64) stays in registers between the two passes -- one round of loads in all
    int32_t x[16];
    int64_t tot = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) { const int i = lo + u * 64 + lane; x[u] = (u * 64 < per && i < hi) ? cnt[i] : 0; tot += (int64_t)x[u]; }
    for (int sh = 1; sh < 64; sh <<= 1) tot += __shfl_xor(tot, sh, 64);
    if (lane == 0) wsum[wave] = tot;
    __syncthreads();
    int64_t carry = 0;
    for (int w = 0; w < wave; w++) carry += wsum[w];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = lo + u * 64 + lane;
      int32_t s = x[u];                                           // (a piece's sum: at most 64 counts, no overflow for any count table of the library)
      for (int sh = 1; sh < 64; sh <<= 1) { const int32_t o = __shfl_up(s, sh, 64); if (lane >= sh) s += o; }
      if (u * 64 < per && i < hi) { ptr[i] = carry + (int64_t)(s - x[u]); cnt[i] = 0; }
      carry += (int64_t)__shfl(s, 63, 64);
    }
    if (threadIdx.x == 1023) ptr[n] = carry;
    return;
  }
  int64_t tot = 0;
  for (int i = lo + lane; i < hi; i += 64) tot += (int64_t)cnt[i];
  for (int sh = 1; sh < 64; sh <<= 1) tot += __shfl_xor(tot, sh, 64);
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  int64_t carry = 0;
  for (int w = 0; w < wave; w++) carry += wsum[w];
  for (int base = lo; base < hi; base += 256) {                  // four 64-entry pieces per trip: their loads are in flight together
    int64_t x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = base + u * 64 + lane; x[u] = i < hi ? (int64_t)cnt[i] : 0; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * 64 + lane;
      int64_t s = x[u];
      for (int sh = 1; sh < 64; sh <<= 1) { const int64_t o = __shfl_up(s, sh, 64); if (lane >= sh) s += o; }
      if (i < hi) { ptr[i] = carry + s - x[u]; cnt[i] = 0; }
      carry += __shfl(s, 63, 64);
    }
  }
  if (threadIdx.x == 1023) ptr[n] = carry;                       // (the last wave's carry after its stretch = the grand total; empty stretches pass it through)
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

// per (row, column slot): base offset into the inverted lists and (len << 16 | V[i,c] bits);
// len = 0 for zero-valued entries.  Lets the row kernel prefetch with one-level loads.
__global__ void colmeta_kernel(const int32_t* __restrict__ q_idx, const hbits* __restrict__ q_val, const int32_t* __restrict__ q_nnz, int capQ,
                               const int64_t* __restrict__ colptr, int row0, int nrows, int32_t* __restrict__ meta_base,
                               uint32_t* __restrict__ meta_lv) {
  const int il = (int)blockIdx.x;
  if (il >= nrows) return;
  const int i = row0 + il, n = q_nnz[i];
  for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x) {
    const hbits v = q_val[(int64_t)i * capQ + p];
    const int c = q_idx[(int64_t)i * capQ + p];
    const int64_t e0 = colptr[c], e1 = colptr[c + 1];
    int len = (v & 0x7fffu) ? (int)(e1 - e0) : 0;
    meta_base[(int64_t)il * capQ + p] = (int32_t)e0;
    meta_lv[(int64_t)il * capQ + p] = ((uint32_t)(len > 0xffff ? 0xffff : len) << 16) | v;   // len >= 65535 handled by the slow path below
  }
}

typedef unsigned int jv4u __attribute__((ext_vector_type(4)));

constexpr int TCAP = 3072;   // touched-column list per wave (uint16 offsets inside the chunk)

// Persistent waves: each wave walks rows il, il+grid, ...  The dense half accumulator t[] stays
// in LDS and is kept all-zero between rows by resetting only the touched entries, so a row costs
// O(nnz) LDS traffic + one streaming constant fill of its J' row + a sparse patch.
__global__ __launch_bounds__(64) void jaccard_rows_kernel(const int32_t* __restrict__ q_nnz, int capQ, const int64_t* __restrict__ colptr,
                                                          const int32_t* __restrict__ q_idx, const int32_t* __restrict__ inv_row,
                                                          const hbits* __restrict__ inv_val, int64_t inv_nnz,
                                                          const int32_t* __restrict__ meta_base, const uint32_t* __restrict__ meta_lv, int N,
                                                          int row0, int nrows, hbits om, hbits* __restrict__ Jp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int cwmax = N < JCHUNK ? N : JCHUNK;
  hbits* t = reinterpret_cast<hbits*>(smem);
  unsigned short* touched = reinterpret_cast<unsigned short*>(smem + (((size_t)cwmax * 2 + 15) & ~(size_t)15) + 16);
  for (int x = lane * 8; x < cwmax; x += 512) *reinterpret_cast<uint4*>(t + x) = make_uint4(0, 0, 0, 0);
  wave_sync2();
  // inverted lists through buffer resources: lanes beyond a list's length read a poisoned
  // offset and get zeros from the bounds check -- no branch around any load
  const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(inv_row), 0, (int)(inv_nnz * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rval = __builtin_amdgcn_make_buffer_rsrc(const_cast<hbits*>(inv_val), 0, (int)(inv_nnz * 2), 0x00020000);
  const hbits jp0 = jaccard_scaled(0, om);   // columns that share nothing with row i
  const unsigned jp0x2 = (unsigned)jp0 | ((unsigned)jp0 << 16);
  const uint64_t lt = lanemask_lt();

  for (int il = (int)blockIdx.x; il < nrows; il += (int)gridDim.x) {
    const int i = row0 + il;
    const int n = q_nnz[i];
    const int32_t* mb = meta_base + (int64_t)il * capQ;
    const uint32_t* ml = meta_lv + (int64_t)il * capQ;
    for (int cbase = 0; cbase < N; cbase += JCHUNK) {
      const int cw = (N - cbase) < JCHUNK ? (N - cbase) : JCHUNK;
      int ntouched = 0;          // > TCAP => overflow, dense epilogue
      // Ascending columns of row i (:110-114), software pipelined: metadata 8 columns ahead,
      // list entries 4 columns ahead of the LDS read-modify-write (named slots, no arrays).
      int mb0, mb1, mb2, mb3, k0, k1, k2, k3, ab0, ab1, ab2, ab3;
      uint32_t ml0, ml1, ml2, ml3, al0, al1, al2, al3;
      hbits v0, v1, v2, v3;
#define SSG_META(P, MB, ML) { const int pp_ = (P) < n ? (P) : (n > 0 ? n - 1 : 0); MB = mb[pp_]; ML = ((P) < n && n > 0) ? ml[pp_] : 0u; }
#define SSG_ENTRIES(MB, ML, K, V, AB, AL)                                                                \
      {                                                                                                 \
        AB = MB; AL = ML;                                                                               \
        const bool in_ = lane < (int)(ML >> 16);                                                        \
        const unsigned o_ = (unsigned)(MB + lane);                                                      \
        K = __builtin_amdgcn_raw_buffer_load_b32(rrow, in_ ? o_ * 4u : 0xfffffff0u, 0, 0);             \
        V = (hbits)__builtin_amdgcn_raw_buffer_load_b16(rval, in_ ? o_ * 2u : 0xfffffff0u, 0, 0);      \
      }
#define SSG_RMW(KK, VV, VIC)                                                                            \
      {                                                                                                 \
        const int kk = (KK) - cbase;                                                                    \
        const bool hit_ = kk >= 0 && kk < cw;                                                           \
        bool first_ = false;                                                                            \
        if (hit_) { const hbits old_ = t[kk]; first_ = (old_ == 0); t[kk] = h_add(old_, h2f(VV) < h2f(VIC) ? (VV) : (VIC)); } \
        const uint64_t fm_ = __ballot(first_);                                                          \
        if (first_) { const int w_ = ntouched + __popcll(fm_ & lt); if (w_ < TCAP) touched[w_] = (unsigned short)kk; } \
        ntouched += __popcll(fm_);                                                                      \
      }
#define SSG_APPLY(K, V, AB, AL)                                                                         \
      {                                                                                                 \
        const int len_ = (int)(AL >> 16);                                                               \
        const hbits vic_ = (hbits)(AL & 0xffffu);                                                       \
        if (len_ > 0) {                                                                                 \
          { const int kq_ = lane < len_ ? K : -1; SSG_RMW(kq_ < 0 ? cbase - 1 : kq_, V, vic_) }         \
          if (len_ > 64) {                                                                              \
            const int c_ = q_idx[(int64_t)i * capQ + pcur_];                                            \
            const int64_t e1_ = colptr[c_ + 1];                                                         \
            for (int64_t eb = (int64_t)AB + 64; eb < e1_; eb += 64) {                                   \
              const int64_t e = eb + lane;                                                              \
              const int kr_ = e < e1_ ? inv_row[e] : cbase - 1;                                         \
              const hbits vk_ = e < e1_ ? inv_val[e] : (hbits)0;                                        \
              SSG_RMW(kr_, vk_, vic_)                                                                   \
            }                                                                                           \
          }                                                                                             \
          wave_sync2();                                                                                 \
        }                                                                                               \
      }
      SSG_META(0, mb0, ml0) SSG_META(1, mb1, ml1) SSG_META(2, mb2, ml2) SSG_META(3, mb3, ml3)
      SSG_ENTRIES(mb0, ml0, k0, v0, ab0, al0) SSG_ENTRIES(mb1, ml1, k1, v1, ab1, al1)
      SSG_ENTRIES(mb2, ml2, k2, v2, ab2, al2) SSG_ENTRIES(mb3, ml3, k3, v3, ab3, al3)
      SSG_META(4, mb0, ml0) SSG_META(5, mb1, ml1) SSG_META(6, mb2, ml2) SSG_META(7, mb3, ml3)
      for (int p = 0; p < n; p += 4) {
        int pcur_ = p;
        SSG_APPLY(k0, v0, ab0, al0) SSG_ENTRIES(mb0, ml0, k0, v0, ab0, al0) SSG_META(p + 8, mb0, ml0)
        pcur_ = p + 1;
        SSG_APPLY(k1, v1, ab1, al1) SSG_ENTRIES(mb1, ml1, k1, v1, ab1, al1) SSG_META(p + 9, mb1, ml1)
        pcur_ = p + 2;
        SSG_APPLY(k2, v2, ab2, al2) SSG_ENTRIES(mb2, ml2, k2, v2, ab2, al2) SSG_META(p + 10, mb2, ml2)
        pcur_ = p + 3;
        SSG_APPLY(k3, v3, ab3, al3) SSG_ENTRIES(mb3, ml3, k3, v3, ab3, al3) SSG_META(p + 11, mb3, ml3)
      }
#undef SSG_META
#undef SSG_ENTRIES
#undef SSG_RMW
#undef SSG_APPLY
      hbits* out = Jp + (int64_t)il * N + cbase;
      const int64_t eoff = (int64_t)il * N + cbase;
      if (ntouched <= TCAP) {
        // (1) stream the constant over the whole chunk, (2) wait for the stores, (3) patch the touched columns
        const int head = (int)((8 - (eoff & 7)) & 7);          // scalar elements before the first 16-byte boundary
        for (int x = lane; x < head && x < cw; x += 64) out[x] = jp0;
        const int nvec = cw > head ? (cw - head) / 8 : 0;
        for (int q = lane; q < nvec; q += 64) *reinterpret_cast<uint4*>(out + head + q * 8) = make_uint4(jp0x2, jp0x2, jp0x2, jp0x2);
        for (int x = head + nvec * 8 + lane; x < cw; x += 64) out[x] = jp0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int q = lane; q < ntouched; q += 64) { const int kk = touched[q]; out[kk] = jaccard_scaled(t[kk], om); t[kk] = 0; }
      } else {
        for (int x = lane; x < cw; x += 64) { const hbits tv = t[x]; out[x] = tv ? jaccard_scaled(tv, om) : jp0; t[x] = 0; }
      }
      wave_sync2();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- round 4
// Second generation of the row kernel.  Same sparse walk (the reference's sequential half adds, ascending columns), new epilogue:
//   * the touched columns are finished IN LDS first (t[k] <- J'[i,k] | 0x8000: a J' value is a non-negative half, so bit 15 is
//     free to mark "touched"), then ONE streaming pass turns the LDS row into the J' row -- marked halves give their value, all
//     others the constant J'(0) -- with whole 16-byte stores, and zeroes the LDS row behind itself.  Every line of J' is written
//     exactly once (the first generation wrote the constant, waited for the stores, then patched 2-byte pieces: 577 MB of HBM
//     writes for 512, and a full drain of the store queue per row);
//   * the touched entries also go out as a SPARSE copy S of the row: packed (J' << 17 | column) words in a pool, one segment per
//     (row, column chunk).  Everything not in S equals the constant J'(0) = half(1 - lambda), the largest value a row holds; the
//     eps rule and the region query only ever ask for entries BELOW a bound, so while that bound stays under J'(0) they walk S
//     (a few hundred entries per row) instead of streaming the N columns again (cluster.hip);
//   * no store drain between rows, and the next row's list heads are fetched before the epilogue of the current one, so that a
//     wave's rows overlap (the first generation paid the q_nnz -> metadata -> entries load chain at the start of every row).
struct SparseOut {
  uint32_t* pool;                      // packed touched entries
  unsigned long long cap;              // pool capacity in entries
  unsigned long long* cursor;          // [0] = entries allocated so far, [1] = 1 when a segment did not fit (S is then unusable)
  int64_t* seg_off;                    // [nrows * nseg] start of the segment in the pool (-1: dropped)
  int32_t* seg_len;                    // [nrows * nseg]
  int nseg;                            // column chunks per row
};

__global__ __launch_bounds__(64) void jaccard_rows2_kernel(const int32_t* __restrict__ q_nnz, int capQ, const int64_t* __restrict__ colptr,
                                                           const int32_t* __restrict__ q_idx, const int32_t* __restrict__ inv_row,
                                                           const hbits* __restrict__ inv_val, int64_t inv_nnz,
                                                           const int32_t* __restrict__ meta_base, const uint32_t* __restrict__ meta_lv, int N,
                                                           int row0, int nrows, hbits om, hbits* __restrict__ Jp, SparseOut so) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int cwmax = N < JCHUNK ? N : JCHUNK;
  hbits* t = reinterpret_cast<hbits*>(smem);
  const int tpad = (cwmax + 7) & ~7;         // the LDS row is read in 8-column vectors
  unsigned short* touched = reinterpret_cast<unsigned short*>(smem + (((size_t)tpad * 2 + 15) & ~(size_t)15) + 16);
  for (int x = lane * 8; x < tpad; x += 512) *reinterpret_cast<uint4*>(t + x) = make_uint4(0, 0, 0, 0);
  wave_sync2();
  const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(inv_row), 0, (int)(inv_nnz * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rval = __builtin_amdgcn_make_buffer_rsrc(const_cast<hbits*>(inv_val), 0, (int)(inv_nnz * 2), 0x00020000);
  const hbits jp0 = jaccard_scaled(0, om);   // columns that share nothing with row i
  const unsigned jp0x2 = (unsigned)jp0 | ((unsigned)jp0 << 16);
  const uint64_t lt = lanemask_lt();

  // walk state of one (row, chunk): list heads four columns deep + metadata eight columns deep (named slots, no arrays)
  int n = 0, mb0, mb1, mb2, mb3, k0, k1, k2, k3, ab0, ab1, ab2, ab3;
  uint32_t ml0, ml1, ml2, ml3, al0, al1, al2, al3;
  hbits v0, v1, v2, v3;
  const int32_t* mb = meta_base;
  const uint32_t* ml = meta_lv;
#define SSG_META(P, MB, ML) { const int pp_ = (P) < n ? (P) : (n > 0 ? n - 1 : 0); MB = mb[pp_]; ML = ((P) < n && n > 0) ? ml[pp_] : 0u; }
#define SSG_ENTRIES(MB, ML, K, V, AB, AL)                                                                \
  {                                                                                                     \
    AB = MB; AL = ML;                                                                                   \
    const bool in_ = lane < (int)(ML >> 16);                                                            \
    const unsigned o_ = (unsigned)(MB + lane);                                                          \
    K = __builtin_amdgcn_raw_buffer_load_b32(rrow, in_ ? o_ * 4u : 0xfffffff0u, 0, 0);                 \
    V = (hbits)__builtin_amdgcn_raw_buffer_load_b16(rval, in_ ? o_ * 2u : 0xfffffff0u, 0, 0);          \
  }
#define SSG_PROLOGUE(IL)                                                                                \
  {                                                                                                     \
    n = q_nnz[row0 + (IL)];                                                                             \
    mb = meta_base + (int64_t)(IL) * capQ; ml = meta_lv + (int64_t)(IL) * capQ;                         \
    SSG_META(0, mb0, ml0) SSG_META(1, mb1, ml1) SSG_META(2, mb2, ml2) SSG_META(3, mb3, ml3)             \
    SSG_ENTRIES(mb0, ml0, k0, v0, ab0, al0) SSG_ENTRIES(mb1, ml1, k1, v1, ab1, al1)                     \
    SSG_ENTRIES(mb2, ml2, k2, v2, ab2, al2) SSG_ENTRIES(mb3, ml3, k3, v3, ab3, al3)                     \
    SSG_META(4, mb0, ml0) SSG_META(5, mb1, ml1) SSG_META(6, mb2, ml2) SSG_META(7, mb3, ml3)             \
  }
  int il = (int)blockIdx.x, cbase = 0;
  if (il < nrows) SSG_PROLOGUE(il)
  while (il < nrows) {
    const int i = row0 + il;
    const int cw = (N - cbase) < JCHUNK ? (N - cbase) : JCHUNK;
    int ntouched = 0;          // > TCAP => the list overflowed, dense patch pass
#define SSG_RMW(KK, VV, VIC)                                                                            \
    {                                                                                                   \
      const int kk = (KK) - cbase;                                                                      \
      const bool hit_ = kk >= 0 && kk < cw;                                                             \
      bool first_ = false;                                                                              \
      if (hit_) { const hbits old_ = t[kk]; first_ = (old_ == 0); t[kk] = h_add(old_, h2f(VV) < h2f(VIC) ? (VV) : (VIC)); } \
      const uint64_t fm_ = __ballot(first_);                                                            \
      if (first_) { const int w_ = ntouched + __popcll(fm_ & lt); if (w_ < TCAP) touched[w_] = (unsigned short)kk; } \
      ntouched += __popcll(fm_);                                                                        \
    }
#define SSG_APPLY(K, V, AB, AL)                                                                         \
    {                                                                                                   \
      const int len_ = (int)(AL >> 16);                                                                 \
      const hbits vic_ = (hbits)(AL & 0xffffu);                                                         \
      if (len_ > 0) {                                                                                   \
        { const int kq_ = lane < len_ ? K : -1; SSG_RMW(kq_ < 0 ? cbase - 1 : kq_, V, vic_) }           \
        if (len_ > 64) {                                                                                \
          const int c_ = q_idx[(int64_t)i * capQ + pcur_];                                              \
          const int64_t e1_ = colptr[c_ + 1];                                                           \
          for (int64_t eb = (int64_t)AB + 64; eb < e1_; eb += 64) {                                     \
            const int64_t e = eb + lane;                                                                \
            const int kr_ = e < e1_ ? inv_row[e] : cbase - 1;                                           \
            const hbits vk_ = e < e1_ ? inv_val[e] : (hbits)0;                                          \
            SSG_RMW(kr_, vk_, vic_)                                                                     \
          }                                                                                             \
        }                                                                                               \
        wave_sync2();                                                                                   \
      }                                                                                                 \
    }
    for (int p = 0; p < n; p += 4) {
      int pcur_ = p;
      SSG_APPLY(k0, v0, ab0, al0) SSG_ENTRIES(mb0, ml0, k0, v0, ab0, al0) SSG_META(p + 8, mb0, ml0)
      pcur_ = p + 1;
      SSG_APPLY(k1, v1, ab1, al1) SSG_ENTRIES(mb1, ml1, k1, v1, ab1, al1) SSG_META(p + 9, mb1, ml1)
      pcur_ = p + 2;
      SSG_APPLY(k2, v2, ab2, al2) SSG_ENTRIES(mb2, ml2, k2, v2, ab2, al2) SSG_META(p + 10, mb2, ml2)
      pcur_ = p + 3;
      SSG_APPLY(k3, v3, ab3, al3) SSG_ENTRIES(mb3, ml3, k3, v3, ab3, al3) SSG_META(p + 11, mb3, ml3)
    }
#undef SSG_RMW
#undef SSG_APPLY
    // the walk of this (row, chunk) is over, its slots are dead: fetch the list heads of the next one now, under the epilogue below
    const int il_cur = il, cbase_cur = cbase;
    if (cbase + JCHUNK < N) cbase += JCHUNK; else { cbase = 0; il += (int)gridDim.x; }
    if (il < nrows) SSG_PROLOGUE(il)

    // ---- patch pass: finish the touched columns in LDS (value | 0x8000) and emit them as the sparse segment of this (row, chunk)
    const int seg = il_cur * so.nseg + cbase_cur / JCHUNK;
    unsigned long long sbase = 0;
    bool sok = false;
    if (so.pool && ntouched > 0) {
      if (lane == 0) sbase = atomicAdd(so.cursor, (unsigned long long)ntouched);
      sbase = (unsigned long long)__shfl((long long)sbase, 0, 64);
      sok = sbase + (unsigned long long)ntouched <= so.cap;
      if (!sok && lane == 0) so.cursor[1] = 1ull;
    }
    if (so.pool && lane == 0) { so.seg_off[seg] = (ntouched == 0 || sok) ? (int64_t)sbase : -1; so.seg_len[seg] = ntouched; }
    if (ntouched <= TCAP) {
      for (int q = lane; q < ntouched; q += 64) {
        const int kk = touched[q];
        const hbits jp = jaccard_scaled(t[kk], om);
        t[kk] = (hbits)(jp | 0x8000u);
        if (sok) so.pool[sbase + q] = ((uint32_t)jp << 17) | (uint32_t)(cbase_cur + kk);
      }
    } else {
      int run = 0;
      for (int x0 = 0; x0 < cw; x0 += 64) {
        const int x = x0 + lane;
        const hbits tv = x < cw ? t[x] : (hbits)0;
        const bool nz = tv != 0;
        hbits jp = 0;
        if (nz) { jp = jaccard_scaled(tv, om); t[x] = (hbits)(jp | 0x8000u); }
        const uint64_t bm = __ballot(nz);
        if (nz && sok) so.pool[sbase + run + __popcll(bm & lt)] = ((uint32_t)jp << 17) | (uint32_t)(cbase_cur + x);
        run += __popcll(bm);
      }
    }
    wave_sync2();
    // ---- streaming pass: LDS row -> J' row (marked halves: their value, the rest: the constant), LDS row zeroed behind it
    {
      hbits* out = Jp + (int64_t)il_cur * N + cbase_cur;
      const int64_t eoff = (int64_t)il_cur * N + cbase_cur;
      const int head = (int)((8 - (eoff & 7)) & 7);          // scalar elements before the first 16-byte boundary of the output
      auto merge = [&](unsigned w) -> unsigned {
        const unsigned m = ((w >> 15) & 0x00010001u) * 0xffffu;
        return (w & 0x7fff7fffu & m) | (jp0x2 & ~m);
      };
      if (head == 0) {
        const int nvec = cw >> 3;
        // four LDS vectors per lane in flight (one read -> merge -> store per trip would leave every trip waiting on its own LDS read)
        for (int q0 = 0; q0 < nvec; q0 += 256) {
          uint4 x[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int q = q0 + u * 64 + lane; if (q < nvec) x[u] = *reinterpret_cast<const uint4*>(t + q * 8); }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int q = q0 + u * 64 + lane;
            if (q < nvec) {
              *reinterpret_cast<uint4*>(t + q * 8) = make_uint4(0, 0, 0, 0);
#if SSG_JAC_NT_STORE
              { typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
                const v4u_ s_ = {merge(x[u].x), merge(x[u].y), merge(x[u].z), merge(x[u].w)}; __builtin_nontemporal_store(s_, reinterpret_cast<v4u_*>(out + q * 8)); }
#else
              *reinterpret_cast<uint4*>(out + q * 8) = make_uint4(merge(x[u].x), merge(x[u].y), merge(x[u].z), merge(x[u].w));
#endif
            }
          }
        }
        for (int x = (nvec << 3) + lane; x < cw; x += 64) { const hbits tv = t[x]; t[x] = 0; out[x] = (tv & 0x8000u) ? (hbits)(tv & 0x7fffu) : jp0; }
      } else {
        // rows that do not start on a 16-byte boundary (N % 8 != 0): the LDS vectors are re-cut with a funnel shift so that the
        // global stores stay aligned; the scalar head and tail elements are written one by one
        for (int x = lane; x < head && x < cw; x += 64) { const hbits tv = t[x]; out[x] = (tv & 0x8000u) ? (hbits)(tv & 0x7fffu) : jp0; }
        const int nvec = cw > head ? (cw - head) >> 3 : 0;
        for (int q = lane; q < nvec; q += 64) {
          unsigned w[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int x = head + q * 8 + 2 * u;
            w[u] = (unsigned)t[x] | ((unsigned)t[x + 1] << 16);
          }
          *reinterpret_cast<uint4*>(out + head + q * 8) = make_uint4(merge(w[0]), merge(w[1]), merge(w[2]), merge(w[3]));
        }
        for (int x = head + (nvec << 3) + lane; x < cw; x += 64) { const hbits tv = t[x]; out[x] = (tv & 0x8000u) ? (hbits)(tv & 0x7fffu) : jp0; }
        wave_sync2();
        for (int x = lane * 8; x < tpad; x += 512) *reinterpret_cast<uint4*>(t + x) = make_uint4(0, 0, 0, 0);
      }
    }
    wave_sync2();
  }
#undef SSG_META
#undef SSG_ENTRIES
#undef SSG_PROLOGUE
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

// colmeta: caller workspace of 2 * nrows * capQ int32 (column metadata of this row block)
extern "C" int ssg_jaccard_rows(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int capQ, const int64_t* colptr,
                                const int32_t* inv_row, const uint16_t* inv_val, int64_t inv_nnz, int32_t* colmeta, int N, int row0, int nrows,
                                uint16_t one_minus_lambda_half, uint16_t* Jp, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || inv_nnz < 0 || inv_nnz > 0x3ffffff0LL) { ssg_set_error("ssg_jaccard_rows: bad shape (N=%d nrows=%d nnz=%lld)", N, nrows, (long long)inv_nnz); return SSG_ERR_INVALID; }
  int32_t* meta_base = colmeta;
  uint32_t* meta_lv = reinterpret_cast<uint32_t*>(colmeta + (int64_t)nrows * capQ);
  hipLaunchKernelGGL(colmeta_kernel, dim3(nrows), dim3(64), 0, stream, q_idx, q_val, q_nnz, capQ, colptr, row0, nrows, meta_base, meta_lv);
  const int cw = N < JCHUNK ? N : JCHUNK;
  const size_t lds = (((size_t)cw * 2 + 15) & ~(size_t)15) + 16 + (size_t)TCAP * 2 + 64;
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)jaccard_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int grid = nrows < 256 * per_cu ? nrows : 256 * per_cu;   // persistent waves, one per LDS slot
  hipLaunchKernelGGL(jaccard_rows_kernel, dim3(grid), dim3(64), lds, stream, q_nnz, capQ, colptr, q_idx, inv_row, inv_val, inv_nnz, meta_base, meta_lv,
                     N, row0, nrows, one_minus_lambda_half, Jp);
  SSG_LAUNCH_CHECK("jaccard_rows_kernel");
  return SSG_OK;
}

// Second generation (round 4): the same J' rows, every line written once, plus the sparse copy S of the touched entries.
//   s_pool [s_cap] uint32: packed (J' << 17 | column); seg_off / seg_len [nrows * ssg_jaccard_segments(N)]: the segment of every (row,
//   32768-column chunk) in the pool; s_cursor [2] uint64 (zeroed here): [0] = entries allocated, [1] = 1 when a segment did not fit (S is
//   then unusable and its consumers take the dense passes).  s_pool == NULL: J' only.  Needs 1 - lambda >= 0 (J' values non-negative:
//   bit 15 marks touched columns in LDS); otherwise the first-generation kernel runs and s_cursor[1] is raised.
extern "C" int ssg_jaccard_segments(int N) { return N <= 0 ? 0 : (N + JCHUNK - 1) / JCHUNK; }
extern "C" int ssg_jaccard_rows2(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int capQ, const int64_t* colptr,
                                 const int32_t* inv_row, const uint16_t* inv_val, int64_t inv_nnz, int32_t* colmeta, int N, int row0, int nrows,
                                 uint16_t one_minus_lambda_half, uint16_t* Jp, uint32_t* s_pool, uint64_t s_cap, uint64_t* s_cursor, int64_t* seg_off,
                                 int32_t* seg_len, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || inv_nnz < 0 || inv_nnz > 0x3ffffff0LL || N > (1 << 17)) { ssg_set_error("ssg_jaccard_rows2: bad shape (N=%d nrows=%d nnz=%lld)", N, nrows, (long long)inv_nnz); return SSG_ERR_INVALID; }
  if (s_pool && (!s_cursor || !seg_off || !seg_len)) { ssg_set_error("ssg_jaccard_rows2: sparse output needs s_cursor, seg_off and seg_len"); return SSG_ERR_INVALID; }
  if (s_cursor) SSG_HIP(hipMemsetAsync(s_cursor, 0, 2 * sizeof(uint64_t), stream));
  static int gen = -1;
  if (gen < 0) { const char* e = getenv("SSG_JACCARD_GEN"); gen = e ? atoi(e) : 2; }
  const bool nonneg = (one_minus_lambda_half & 0x8000u) == 0 && (one_minus_lambda_half & 0x7fffu) <= 0x7c00u;
  if (gen != 2 || !nonneg) {
    if (s_cursor) { const uint64_t one = 1; SSG_HIP(hipMemcpyAsync(s_cursor + 1, &one, sizeof(one), hipMemcpyHostToDevice, stream)); }
    return ssg_jaccard_rows(q_idx, q_val, q_nnz, capQ, colptr, inv_row, inv_val, inv_nnz, colmeta, N, row0, nrows, one_minus_lambda_half, Jp, stream);
  }
  int32_t* meta_base = colmeta;
  uint32_t* meta_lv = reinterpret_cast<uint32_t*>(colmeta + (int64_t)nrows * capQ);
  hipLaunchKernelGGL(colmeta_kernel, dim3(nrows), dim3(64), 0, stream, q_idx, q_val, q_nnz, capQ, colptr, row0, nrows, meta_base, meta_lv);
  const int cw = N < JCHUNK ? N : JCHUNK;
  const size_t lds = (((size_t)((cw + 7) & ~7) * 2 + 15) & ~(size_t)15) + 16 + (size_t)TCAP * 2 + 64;
  if (lds > 64 * 1024) SSG_HIP(hipFuncSetAttribute((const void*)jaccard_rows2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int grid = nrows < 256 * per_cu ? nrows : 256 * per_cu;   // persistent waves, one per LDS slot
  SparseOut so; so.pool = s_pool; so.cap = s_cap; so.cursor = (unsigned long long*)s_cursor; so.seg_off = seg_off; so.seg_len = seg_len; so.nseg = ssg_jaccard_segments(N);
  hipLaunchKernelGGL(jaccard_rows2_kernel, dim3(grid), dim3(64), lds, stream, q_nnz, capQ, colptr, q_idx, inv_row, inv_val, inv_nnz, meta_base, meta_lv,
                     N, row0, nrows, one_minus_lambda_half, Jp, so);
  SSG_LAUNCH_CHECK("jaccard_rows2_kernel");
  return SSG_OK;
}

extern "C" int ssg_final_dist_f64(const uint16_t* Jp, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double* out,
                                  hipStream_t stream) {
  if (N <= 0 || nrows <= 0) { ssg_set_error("ssg_final_dist_f64: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(final_dist_kernel, dim3(2048), dim3(256), 0, stream, Jp, v, N, row0, nrows, lambda_value, out);
  SSG_LAUNCH_CHECK("final_dist_kernel");
  return SSG_OK;
}
