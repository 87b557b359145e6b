// topk_intro.hip -- K5 in the UNMODIFIED reference's tie order.
//
// reid/rerank.py:68-70:
//   original_dist = transpose(original_dist / max(original_dist, axis=0))
//   initial_rank  = np.argsort(original_dist).astype(np.int32)
// np.argsort's default kind is numpy's introsort on an index array (npysort/quicksort.cpp
// aquicksort_<half>: median-of-3 Hoare partition, insertion sort below 17 elements, heapsort when the
// depth budget 2*floor(log2 N) runs out).  It is unstable: which of several equal half keys lands in
// column r depends on the whole sequence of partitions, and the normalised half distances tie in
// nearly every row (hard part 1).  This kernel reproduces that order exactly:
//
//   * one workgroup per row; the row's (key, column) pairs live in LDS as packed 32-bit entries
//     (14-bit order key of the normalised half | 18-bit column), or in a global arena when the row
//     does not fit (N > ~36 k);
//   * only the ranges of the quicksort recursion that intersect output columns [0, K) are walked
//     (about 2N element visits per row instead of N log N);
//   * each Hoare partition is executed data-parallel instead of by two sequential scanning
//     pointers: with S = [pl+1, pr-2], L-stoppers = positions whose key is >= the pivot,
//     R-stoppers = positions whose key is <= the pivot, f(p) = #L-stoppers at positions <= p and
//     g(p) = #R-stoppers at positions > p, the scanning pointers perform m = f(p*) swaps, where p* is
//     the last position with g(p) >= f(p); they pair the k-th L-stopper from the left with the k-th
//     R-stopper from the right (k <= m), and the pivot lands at p* + 1.  Stopper bitmasks come from
//     wave ballots, ranks from popcounts and one prefix scan, so a partition costs four barriers.
//     tools/introsort_model.py states the same computation in numpy and is checked against the
//     sequential restatement (oracle/ssg_oracle.c aquicksort_half) on the CPU;
//   * ranges of <= 1024 entries are finished by wave 0 alone (no workgroup barriers).
#include "ssg_common.h"

namespace ssg {
namespace intro {

constexpr int NT = 256;            // threads per workgroup
constexpr int NW = NT / 64;        // waves per workgroup
constexpr int SMALL = 15;          // ranges with pr - pl > SMALL are partitioned (numpy 2.2.6)
constexpr int WAVE_N = 1024;       // ranges up to this many entries are handled by wave 0 alone
constexpr int STACK = 64;          // pending ranges that intersect [0, K): all disjoint with pl < K <= 64
constexpr int IDX_BITS = 18;       // column bits of a packed entry (N <= 262144)
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr uint32_t KEY_NAN = 0x3fffu;

struct Ctl {
  uint64_t wtot[NW];     // stoppers (L low word, R high word) seen by each wave of the flag pass
  uint64_t woff[NW];     // exclusive prefix of wtot
  uint64_t tot;
  int stack[STACK * 3];  // (pl, pr, depth budget) of pushed ranges that still matter
  int sp;
  int wstar;
  uint32_t vp;
};

struct LdsArena {
  uint32_t* p;
  __device__ __forceinline__ uint32_t get(int i) const { return p[i]; }
  __device__ __forceinline__ void set(int i, uint32_t e) const { p[i] = e; }
};
struct GlobalArena {
  uint32_t* p;
  __device__ __forceinline__ uint32_t get(int i) const { return p[i]; }
  __device__ __forceinline__ void set(int i, uint32_t e) const { p[i] = e; }
};

__device__ __forceinline__ uint32_t ekey(uint32_t e) { return e >> IDX_BITS; }
__device__ __forceinline__ uint32_t lo32(uint64_t x) { return (uint32_t)x; }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }

template <bool WAVE>
__device__ __forceinline__ void gsync() {
  if (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// index of the n-th (1-based) set bit of m, counted from bit 0; m must have >= n set bits
__device__ __forceinline__ int nth_set_bit(uint64_t m, int n) {
  int pos = 0;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int c = __popcll((m >> pos) & ((1ull << s) - 1ull));
    if (c < n) { n -= c; pos += s; }
  }
  return pos;
}

struct Masks {
  uint64_t* L;     // bit b of L[w]: position s0 + 64 w + b holds a key >= pivot
  uint64_t* R;     //                                         ...   a key <= pivot
  uint64_t* P;     // inclusive prefix of (popc L | popc R << 32) inside the owning wave's chunk of words
};

// inclusive packed stopper count up to and including word w
__device__ __forceinline__ uint64_t pincl(const Masks& mk, const Ctl* sh, int w, int wpw) { return mk.P[w] + sh->woff[w / wpw]; }

// smallest word whose inclusive count (L: low half, R: high half) reaches r (1-based rank from the left)
template <bool RIGHT>
__device__ __forceinline__ int word_of_rank(const Masks& mk, const Ctl* sh, int W, int wpw, uint32_t r) {
  int lo = 0, hi = W - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const uint64_t p = pincl(mk, sh, mid, wpw);
    const uint32_t c = RIGHT ? hi32(p) : lo32(p);
    if (c >= r) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// One Hoare partition of A[pl..pr] (pr - pl > SMALL) by the whole workgroup (WAVE = false) or by the calling wave
// alone (WAVE = true, W <= 16 words).  Returns the final pivot position.
template <bool WAVE, class Arena>
__device__ __forceinline__ int partition(const Arena& A, Ctl* sh, const Masks& mk, int pl, int pr) {
  const int lane = lane_id();
  const int tid = WAVE ? lane : (int)threadIdx.x;
  const int nthr = WAVE ? 64 : NT;
  const int nwav = WAVE ? 1 : NW;
  const int wav = WAVE ? 0 : (int)(threadIdx.x >> 6);
  const int s0 = pl + 1, s1 = pr - 2;
  const int W = (s1 - s0 + 64) >> 6;
  const int wpw = (W + nwav - 1) / nwav;

  // ---- A: median of three, pivot parked at pr-1 (quicksort.cpp, head of the partition loop)
  if (tid == 0) {
    const int pm = pl + ((pr - pl) >> 1);
    uint32_t el = A.get(pl), em = A.get(pm), er = A.get(pr), t;
    if (ekey(em) < ekey(el)) { t = em; em = el; el = t; }
    if (ekey(er) < ekey(em)) { t = er; er = em; em = t; }
    if (ekey(em) < ekey(el)) { t = em; em = el; el = t; }
    const uint32_t e2 = A.get(pr - 1);
    A.set(pl, el); A.set(pr, er); A.set(pm, e2); A.set(pr - 1, em);
    sh->vp = ekey(em);
    sh->wstar = 0;
  }
  gsync<WAVE>();
  const uint32_t vp = sh->vp;

  // ---- B: stopper masks of the scan region + running counts (each wave owns a contiguous chunk of words)
  {
    const int w0 = wav * wpw, w1 = min(W, w0 + wpw);
    uint64_t run = 0;
    for (int w = w0; w < w1; w++) {
      const int p = s0 + (w << 6) + lane;
      const bool valid = p <= s1;
      const uint32_t k = valid ? ekey(A.get(p)) : 0u;
      const uint64_t lm = __ballot(valid && k >= vp);
      const uint64_t rm = __ballot(valid && k <= vp);
      run += (uint64_t)__popcll(lm) | ((uint64_t)__popcll(rm) << 32);
      if (lane == 0) { mk.L[w] = lm; mk.R[w] = rm; mk.P[w] = run; }
    }
    if (lane == 0) sh->wtot[wav] = run;
  }
  gsync<WAVE>();

  // ---- C: chunk offsets; count the words at whose end g >= f still holds (monotone) -> crossing word
  {
    uint64_t tot = 0;
    for (int u = 0; u < nwav; u++) tot += sh->wtot[u];
    if (tid < nwav) {
      uint64_t off = 0;
      for (int u = 0; u < tid; u++) off += sh->wtot[u];
      sh->woff[tid] = off;
      if (tid == 0) sh->tot = tot;
    }
    const uint32_t totR = hi32(tot);
    int cnt = 0;
    for (int w = tid; w < W; w += nthr) {
      const int c = w / wpw;
      uint64_t off = 0;
      for (int u = 0; u < c; u++) off += sh->wtot[u];
      const uint64_t p = mk.P[w] + off;
      cnt += (totR - hi32(p)) >= lo32(p) ? 1 : 0;
    }
    // wave reduction, one LDS atomic per wave
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && cnt) atomicAdd(&sh->wstar, cnt);
  }
  gsync<WAVE>();

  // ---- D: number of swaps m and the pivot position p* + 1 (every wave computes them redundantly)
  const uint64_t tot = sh->tot;
  const uint32_t totL = lo32(tot), totR = hi32(tot);
  const int wstar = sh->wstar;
  uint32_t m;
  int pstar;
  if (wstar >= W) {
    m = totL; pstar = s1;
  } else {
    const uint64_t lm = mk.L[wstar], rm = mk.R[wstar];
    const uint64_t p = pincl(mk, sh, wstar, wpw);
    const uint32_t cumL = lo32(p) - (uint32_t)__popcll(lm);       // L-stoppers in words < wstar
    const uint32_t cumR = totR - hi32(p);                         // R-stoppers in words > wstar
    const uint64_t le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const uint32_t f = cumL + (uint32_t)__popcll(lm & le);
    const uint32_t g = cumR + (uint32_t)__popcll(rm & ~le);
    const bool valid = s0 + (wstar << 6) + lane <= s1;
    const uint64_t bal = __ballot(valid && g >= f);
    if (bal == 0) {
      m = cumL; pstar = s0 + (wstar << 6) - 1;
    } else {
      const int b = 63 - __clzll((long long)bal);
      const uint64_t leb = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
      m = cumL + (uint32_t)__popcll(lm & leb);
      pstar = s0 + (wstar << 6) + b;
    }
  }
  const int pi = pstar + 1;

  // ---- E: swap the k-th L-stopper from the left with the k-th R-stopper from the right, k <= m
  if (m) {
    const uint32_t c = (m + nthr - 1) / nthr;
    const uint32_t k0 = (uint32_t)tid * c + 1u;
    const uint32_t k1 = min(m, k0 + c - 1u);
    if (k0 <= k1) {
      int wl = word_of_rank<false>(mk, sh, W, wpw, k0);
      uint64_t ml = mk.L[wl];
      {
        const uint32_t before = lo32(pincl(mk, sh, wl, wpw)) - (uint32_t)__popcll(ml);
        const int b = nth_set_bit(ml, (int)(k0 - before));
        ml &= ~((1ull << b) - 1ull);                               // keep bits >= b
      }
      const uint32_t t0 = totR - k0 + 1u;                          // the same stopper counted from the left
      int wr = word_of_rank<true>(mk, sh, W, wpw, t0);
      uint64_t mr = mk.R[wr];
      {
        const uint32_t before = hi32(pincl(mk, sh, wr, wpw)) - (uint32_t)__popcll(mr);
        const int b = nth_set_bit(mr, (int)(t0 - before));
        mr &= (b == 63) ? ~0ull : ((2ull << b) - 1ull);            // keep bits <= b
      }
      for (uint32_t k = k0; k <= k1; k++) {
        while (ml == 0) ml = mk.L[++wl];
        const int bl = __ffsll((long long)ml) - 1;
        ml &= ml - 1;
        while (mr == 0) mr = mk.R[--wr];
        const int br = 63 - __clzll((long long)mr);
        mr &= ~(1ull << br);
        const int a = s0 + (wl << 6) + bl, b = s0 + (wr << 6) + br;
        const uint32_t ea = A.get(a), eb = A.get(b);
        A.set(a, eb); A.set(b, ea);
      }
    }
  }
  gsync<WAVE>();

  // ---- F: pivot into place (visible to thread 0 / wave 0, which own every step that follows before the next barrier)
  if (tid == 0) {
    const uint32_t e1 = A.get(pi), e2 = A.get(pr - 1);
    A.set(pi, e2); A.set(pr - 1, e1);
  }
  return pi;
}

// heapsort.cpp aheapsort_ on A[lo .. lo+n) (one thread; only reached past the depth budget)
template <class Arena>
__device__ __forceinline__ void heapsort(const Arena& A, int lo, int n) {
  const int base = lo - 1;   // 1-based heap
  int i, j, l;
  uint32_t tmp;
  for (l = n >> 1; l > 0; --l) {
    tmp = A.get(base + l);
    for (i = l, j = l << 1; j <= n;) {
      if (j < n && ekey(A.get(base + j)) < ekey(A.get(base + j + 1))) j += 1;
      if (ekey(tmp) < ekey(A.get(base + j))) { A.set(base + i, A.get(base + j)); i = j; j += j; } else break;
    }
    A.set(base + i, tmp);
  }
  for (; n > 1;) {
    tmp = A.get(base + n); A.set(base + n, A.get(base + 1)); n -= 1;
    for (i = 1, j = 2; j <= n;) {
      if (j < n && ekey(A.get(base + j)) < ekey(A.get(base + j + 1))) j++;
      if (ekey(tmp) < ekey(A.get(base + j))) { A.set(base + i, A.get(base + j)); i = j; j += j; } else break;
    }
    A.set(base + i, tmp);
  }
}

// insertion sort of A[pl..pr] (<= 16 entries) == stable rank sort, one wave
template <class Arena>
__device__ __forceinline__ void insertion(const Arena& A, int pl, int pr) {
  gsync<true>();
  const int lane = lane_id(), n = pr - pl + 1;
  const uint32_t e = lane < n ? A.get(pl + lane) : 0xffffffffu;
  const uint32_t k = ekey(e);
  int r = 0;
  for (int q = 0; q < n; q++) {
    const uint32_t kq = (uint32_t)__shfl((int)k, q);
    r += (kq < k || (kq == k && q < lane)) ? 1 : 0;
  }
  gsync<true>();
  if (lane < n) A.set(pl + r, e);
}

// children of a partition: which one the quicksort loop continues with, which one it pushes
struct Split { int cl, cr, ql, qr; };
__device__ __forceinline__ Split split_ranges(int pl, int pr, int pi) {
  Split s;
  if (pi - pl < pr - pi) { s.cl = pl; s.cr = pi - 1; s.ql = pi + 1; s.qr = pr; }
  else { s.cl = pi + 1; s.cr = pr; s.ql = pl; s.qr = pi - 1; }
  return s;
}

// sorts exactly the ranges of the introsort recursion that intersect [0, K)
template <class Arena>
__device__ __forceinline__ void sort_prefix(const Arena& A, Ctl* sh, const Masks& mk, int N, int K) {
  const int tid = (int)threadIdx.x, lane = lane_id(), wav = tid >> 6;
  int sp = 0;
  int pl = 0, pr = N - 1, cd = 2 * (31 - __clz(N));
  bool have = true;
  for (;;) {
    if (!have) {
      if (sp == 0) break;
      __syncthreads();
      --sp;
      pl = sh->stack[3 * sp]; pr = sh->stack[3 * sp + 1]; cd = sh->stack[3 * sp + 2];
      if (cd < 0) {                       // popped past the depth budget: heapsort the whole range
        if (tid == 0) heapsort(A, pl, pr - pl + 1);
        continue;
      }
    }
    have = false;
    bool needed = true;
    while (pr - pl > SMALL && pr - pl + 1 > WAVE_N) {
      const int pi = partition<false>(A, sh, mk, pl, pr);
      --cd;
      const Split s = split_ranges(pl, pr, pi);
      if (s.ql < K && s.ql <= s.qr) {
        if (tid == 0) { sh->stack[3 * sp] = s.ql; sh->stack[3 * sp + 1] = s.qr; sh->stack[3 * sp + 2] = cd; }
        ++sp;
      }
      if (s.cl < K && s.cl <= s.cr) { pl = s.cl; pr = s.cr; }
      else { needed = false; break; }
    }
    if (needed) {
      if (wav == 0) {
        int wsp = sp;
        while (pr - pl > SMALL) {
          const int pi = partition<true>(A, sh, mk, pl, pr);
          --cd;
          const Split s = split_ranges(pl, pr, pi);
          if (s.ql < K && s.ql <= s.qr) {
            if (lane == 0) { sh->stack[3 * wsp] = s.ql; sh->stack[3 * wsp + 1] = s.qr; sh->stack[3 * wsp + 2] = cd; }
            ++wsp;
          }
          if (s.cl < K && s.cl <= s.cr) { pl = s.cl; pr = s.cr; }
          else { needed = false; break; }
        }
        if (needed && pr > pl) insertion(A, pl, pr);
        if (lane == 0) sh->sp = wsp;
      }
      __syncthreads();
      sp = sh->sp;
    }
  }
  __syncthreads();
}

// order key of half(raw / rowmax): the half bit pattern (values in [0, 1]); NaN sorts last like numpy's half less-than
__device__ __forceinline__ uint32_t norm_key(hbits raw, float fmx) {
  const hbits q = f2h(h2f(raw) / fmx);
  if (h_isnan(q)) return KEY_NAN;
  return q > 0x3ffeu ? 0x3ffeu : (uint32_t)q;
}

template <bool LDS>
__global__ __launch_bounds__(NT) void topk_introsort_kernel(const hbits* __restrict__ D, const unsigned* __restrict__ rowmax, int N, int nrows,
                                                            int K, int wcap, uint32_t* __restrict__ arena, int32_t* __restrict__ rank) {
  extern __shared__ __align__(16) unsigned char smem[];
  Ctl* sh = reinterpret_cast<Ctl*>(smem);
  Masks mk;
  mk.L = reinterpret_cast<uint64_t*>(smem + ((sizeof(Ctl) + 15) & ~(size_t)15));
  mk.R = mk.L + wcap;
  mk.P = mk.R + wcap;
  uint32_t* ent = reinterpret_cast<uint32_t*>(mk.P + wcap);
  const int tid = (int)threadIdx.x;
  for (int row = (int)blockIdx.x; row < nrows; row += (int)gridDim.x) {
    const float fmx = h2f((hbits)rowmax[row]);
    // ---- the row as packed (key, column) entries
    const int64_t total = (int64_t)nrows * N;
    const int64_t base = (int64_t)row * N;
    const int64_t al = base & ~(int64_t)7;
    const int first = (int)(base - al);
    const int nch = (first + N + 7) >> 3;
    uint32_t* dst = LDS ? ent : arena + (size_t)blockIdx.x * (size_t)N;
    for (int c = tid; c < nch; c += NT) {
      const int64_t off = al + (int64_t)c * 8;
      unsigned w[4] = {0, 0, 0, 0};
      if (off + 8 <= total) {
        const uint4 x = *reinterpret_cast<const uint4*>(D + off);
        w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (off + e < total) w[e >> 1] |= (unsigned)D[off + e] << ((e & 1) * 16);
      }
      const int j0 = c * 8 - first;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int j = j0 + e;
        if (j >= 0 && j < N) {
          const hbits r = (hbits)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          const uint32_t ev = (norm_key(r, fmx) << IDX_BITS) | (uint32_t)j;
          dst[j] = ev;
        }
      }
    }
    __syncthreads();
    if (LDS) {
      LdsArena A{ent};
      sort_prefix(A, sh, mk, N, K);
      if (tid < K) rank[(int64_t)row * K + tid] = (int32_t)(A.get(tid) & IDX_MASK);
    } else {
      GlobalArena A{dst};
      sort_prefix(A, sh, mk, N, K);
      if (tid < K) rank[(int64_t)row * K + tid] = (int32_t)(A.get(tid) & IDX_MASK);
    }
    __syncthreads();
  }
}

constexpr size_t LDS_LIMIT = 160 * 1024;
__host__ inline int mask_words(int N) { return (N + 63) / 64 + 1; }
__host__ inline size_t lds_fixed_bytes(int N) { return ((sizeof(Ctl) + 15) & ~(size_t)15) + (size_t)mask_words(N) * 3 * sizeof(uint64_t); }
__host__ inline bool fits_lds(int N) { return lds_fixed_bytes(N) + (size_t)N * 4 <= LDS_LIMIT; }
__host__ inline int arena_blocks(int nrows) { return nrows < 2048 ? nrows : 2048; }

}  // namespace intro
}  // namespace ssg

using namespace ssg;

extern "C" size_t ssg_topk_rank_introsort_arena_bytes(int N, int nrows) {
  if (N <= 0 || nrows <= 0) return 0;
  return (size_t)intro::arena_blocks(nrows) * (size_t)N * sizeof(uint32_t);
}
extern "C" size_t ssg_topk_rank_introsort_ws_bytes(int N, int nrows) {
  return (N > 0 && intro::fits_lds(N)) ? 0 : ssg_topk_rank_introsort_arena_bytes(N, nrows);
}

extern "C" int ssg_topk_rank_introsort(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, void* ws,
                                       size_t ws_bytes, hipStream_t stream) {
  if (N < 2 || nrows <= 0 || K <= 0 || K > 64 || K > N || N > (1 << intro::IDX_BITS)) {
    ssg_set_error("ssg_topk_rank_introsort: need 0 < K <= min(64, N), 2 <= N <= 262144 (K=%d N=%d)", K, N);
    return SSG_ERR_INVALID;
  }
  const int wcap = intro::mask_words(N);
  const bool arena = !intro::fits_lds(N) || (ws != nullptr && ws_bytes > 0 && ws_bytes >= ssg_topk_rank_introsort_arena_bytes(N, nrows));
  if (!arena) {
    const size_t lds = intro::lds_fixed_bytes(N) + (size_t)N * 4;
    SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&intro::topk_introsort_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(intro::topk_introsort_kernel<true>, dim3(nrows), dim3(intro::NT), lds, stream, D, rowmax, N, nrows, K, wcap,
                       (uint32_t*)nullptr, rank);
  } else {
    const size_t need = ssg_topk_rank_introsort_arena_bytes(N, nrows);
    if (ws == nullptr || ws_bytes < need) {
      ssg_set_error("ssg_topk_rank_introsort: workspace of %zu bytes needed for N=%d (got %zu)", need, N, ws_bytes);
      return SSG_ERR_INVALID;
    }
    const size_t lds = intro::lds_fixed_bytes(N);
    SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&intro::topk_introsort_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(intro::topk_introsort_kernel<false>, dim3(intro::arena_blocks(nrows)), dim3(intro::NT), lds, stream, D, rowmax, N,
                       nrows, K, wcap, (uint32_t*)ws, rank);
  }
  SSG_LAUNCH_CHECK("topk_introsort_kernel");
  return SSG_OK;
}
