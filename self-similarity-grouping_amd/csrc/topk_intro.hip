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
//   * one workgroup per row; the row lives in LDS as packed 32-bit entries (15-bit raw half distance |
//     17-bit column), or in a global arena when it does not fit (N > ~36 k).  The sort key is
//     key(raw) = half(raw / rowmax), a monotone step function of the raw value, so a comparison with a
//     pivot key is a comparison of the raw value with the two ends of the pivot's key class
//     [lo, hi] = {raw : key(raw) == key(pivot)}: only the pivots (and the <= 16-entry tails) are ever
//     divided, not the N entries;
//   * only the ranges of the quicksort recursion that intersect output columns [0, K) are walked
//     (about 2N element visits per row instead of N log N);
//   * each Hoare partition is executed data-parallel instead of by two sequential scanning
//     pointers: with S = [pl+1, pr-2], L-stoppers = positions whose key is >= the pivot,
//     R-stoppers = positions whose key is <= the pivot, f(p) = #L-stoppers at positions <= p and
//     g(p) = #R-stoppers at positions > p, the scanning pointers perform m = f(p*) swaps, where p* is
//     the last position with g(p) >= f(p); they pair the k-th L-stopper from the left with the k-th
//     R-stopper from the right (k <= m), and the pivot lands at p* + 1.  Stopper bitmasks come from
//     wave-wide compares, ranks from popcounts and one DPP prefix scan; a partition costs three barriers.
//     tools/introsort_model.py states the same computation in numpy and is checked against the
//     sequential restatement (oracle/ssg_oracle.c aquicksort_half) on the CPU;
//   * ranges of <= 1024 entries are finished by wave 0 alone (no workgroup barriers).
#include "ssg_common.h"
#include <algorithm>
#include <cstdlib>

namespace ssg {
namespace intro {

#ifdef SSG_INTRO_PROF
// cycle accounting for tools/micro/intro_prof.hip: thread 0 accumulates s_memtime deltas per phase in registers
// (compile-time slots) and publishes them once per row
__device__ unsigned long long g_prof[16];
struct ProfAcc { unsigned long long a[16]; };
#define PROF_ARG , ProfAcc& pacc_
#define PROF_PASS , pacc_
#define PROF_DECL unsigned long long prof_t_ = clock64()
#define PROF(slot) do { const unsigned long long n_ = clock64(); pacc_.a[slot] += n_ - prof_t_; prof_t_ = n_; } while (0)
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_DECL do {} while (0)
#define PROF(slot) do {} while (0)
#endif

// ranges with pr - pl > SMALL are partitioned.  numpy-build dependent: the numpy 2.2.6 wheel of this image (and the goldens
// generated with it) behaves like 15 -- 17 elements are still partitioned -- while numpy's source constant reads 16;
// tests/test_oracle_golden.py::test_installed_numpy_introsort_threshold probes the installed numpy.  Build with
// -DSSG_INTRO_SMALL=16 for a numpy that follows the source constant (the insertion-sort tail then holds up to 17 entries).
#ifndef SSG_INTRO_SMALL
#define SSG_INTRO_SMALL 15
#endif
constexpr int SMALL = SSG_INTRO_SMALL;
#ifndef SSG_INTRO_WAVE_N
#define SSG_INTRO_WAVE_N 1024
#endif
constexpr int WAVE_N = SSG_INTRO_WAVE_N;   // ranges up to this many entries are handled by wave 0 alone
constexpr int STACK = 64;          // pending ranges that intersect [0, K): all disjoint with pl < K <= 64
constexpr int IDX_BITS = 17;       // column bits of a packed entry (N <= 131072); the raw half (< 0x8000) sits above
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr uint32_t KEY_NAN = 0xffffu;
constexpr int MAXCH = 64;          // chunks of mask words per partition (one lane each in the chunk scan)

struct Ctl {
  uint32_t ctotL[MAXCH], ctotR[MAXCH];   // stoppers per chunk of 2^cs mask words
  int stack[STACK * 3];                  // (pl, pr, depth budget) of pushed ranges that still matter
  int sp;
  uint32_t xp;                           // raw half of the current pivot
  int ntail;                             // split mode: ranges handed to the tail kernel (they stay on `tail`, same triples)
  int tail[STACK * 3];
  uint32_t ov_el, ov_e2;                 // streamed partition: what the median-of-3 step leaves at positions 0 and pm of the (read-only) source
};

// Split mode (default): the workgroup kernel only runs the partitions of ranges longer than `tailn` entries and hands every
// shorter range that still intersects [0, K) to the tail kernel, where ONE WAVE per row finishes them -- 8 to 25 rows per CU at a
// time instead of one wave working while the seven others of the row's workgroup idle (a third of a row's time in round 2).
// Hand-over record of a row in global memory: TailHdr, then the entries A[0 .. hi) (hi <= K + tailn).
struct TailHdr { int n, hi, pad0, pad1; int rng[STACK * 3]; };
constexpr int TAIL_HDR_WORDS = (int)(sizeof(TailHdr) / 4);
static_assert(TAIL_HDR_WORDS % 4 == 0, "the entries behind the header start on a 16-byte boundary");
constexpr int NOCHECK = 1 << 20;   // added to the depth budget of a handed-over range that must not be depth-tested when the tail kernel pops it

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

__device__ __forceinline__ uint32_t eraw(uint32_t e) { return e >> IDX_BITS; }

// order key of half(raw / rowmax): the half bit pattern (values in [0, 1]); NaN sorts last like numpy's half less-than
__device__ __forceinline__ uint32_t norm_key(uint32_t raw, float fmx) {
  const hbits q = f2h(h2f((hbits)raw) / fmx);
  return h_isnan(q) ? KEY_NAN : (uint32_t)q;
}

// values that are wave-uniform by construction but live in VGPRs (loaded from LDS, shuffled): moving them to SGPRs lets the
// compiler run loop control and 64-bit mask arithmetic on the scalar unit instead of issuing them for 64 lanes
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni(uint64_t x) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}

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
  uint32_t* P;     // inclusive stopper counts from the start of the word's chunk up to and including word w: L | R << 16
};

// wave64 inclusive prefix sum on the DPP network (row shifts inside the 16-lane rows, then row broadcasts)
template <int CTRL, int ROWM, int BANKM, bool BC>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROWM, BANKM, BC);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  uint32_t v = x + dpp_mov<0x111, 0xf, 0xf, true>(x);     // row_shr:1
  v += dpp_mov<0x112, 0xf, 0xf, true>(x);                 // row_shr:2
  v += dpp_mov<0x113, 0xf, 0xf, true>(x);                 // row_shr:3
  v += dpp_mov<0x114, 0xf, 0xe, false>(v);                // row_shr:4, banks 1-3
  v += dpp_mov<0x118, 0xf, 0xc, false>(v);                // row_shr:8, banks 2-3
  v += dpp_mov<0x142, 0xa, 0xf, false>(v);                // row_bcast:15 -> rows 1, 3
  v += dpp_mov<0x143, 0xc, 0xf, false>(v);                // row_bcast:31 -> rows 2, 3
  return v;
}

// word holding the r-th (1-based, from the left) stopper: chunk through the per-lane chunk prefix `ex` (lane c = stoppers in
// chunks < c; nchp = power of two >= number of chunks), then the word inside the chunk through its running counts P
template <bool RIGHT>
__device__ __forceinline__ int word_of_rank(const uint32_t* P, uint32_t ex, int W, int cs, int nch, int nchp, uint32_t r, uint32_t& before_chunk) {
  int lo = 0;                            // last chunk whose exclusive offset is < r (ex[0] = 0 < r); wave-uniform trip count: the
  for (int step = nchp >> 1; step >= 1; step >>= 1) {      // shuffles need every lane of the wave active
    const int cand = lo + step;
    const uint32_t v = (uint32_t)__shfl((int)ex, cand & 63);
    if (cand < nch && v < r) lo = cand;
  }
  const uint32_t off = (uint32_t)__shfl((int)ex, lo);
  before_chunk = off;
  int wl = lo << cs, wh = min(W, wl + (1 << cs)) - 1;
  const uint32_t rr = r - off;
  while (wl < wh) {
    const int mid = (wl + wh) >> 1;
    const uint32_t c = RIGHT ? (P[mid] >> 16) : (P[mid] & 0xffffu);
    if (c >= rr) wh = mid; else wl = mid + 1;
  }
  return wl;
}

// ends [lo, hi] of the raw values that share the pivot's key (key is monotone in raw): the lanes probe 64 neighbours at a
// time (one or two raw values share a key in the normal range; arbitrarily many when the key is a half subnormal)
__device__ __forceinline__ void key_class(uint32_t xp, float fmx, uint32_t& lo, uint32_t& hi) {
  const uint32_t lane1 = (uint32_t)lane_id() + 1u;
  const uint32_t kp = norm_key(xp, fmx);
  lo = xp; hi = xp;
  for (;;) {
    const bool same = lo >= lane1 && norm_key(lo - lane1, fmx) == kp;
    const uint64_t b = __ballot(same);
    const int n = (b == ~0ull) ? 64 : __builtin_ctzll(~b);
    lo -= (uint32_t)n;
    if (n < 64) break;
  }
  for (;;) {
    const bool same = hi + lane1 <= 0x7fffu && norm_key(hi + lane1, fmx) == kp;
    const uint64_t b = __ballot(same);
    const int n = (b == ~0ull) ? 64 : __builtin_ctzll(~b);
    hi += (uint32_t)n;
    if (n < 64) break;
  }
}

// One Hoare partition of A[pl..pr] (pr - pl > SMALL) by the whole workgroup (WAVE = false) or by the calling wave
// alone (WAVE = true).  Returns the final pivot position.  Three barriers: A | B | C+D+E | (F).
template <bool WAVE, int NT, class Arena>
__device__ __forceinline__ int partition(const Arena& A, Ctl* sh, const Masks& mk, float fmx, int pl, int pr PROF_ARG) {
  constexpr int NW = NT / 64;
  pl = uni(pl); pr = uni(pr);
  const int lane = lane_id();
  const int tid = WAVE ? lane : (int)threadIdx.x;
  constexpr int nthr = WAVE ? 64 : NT;
  constexpr int nwav = WAVE ? 1 : NW;
  const int wav = WAVE ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int s0 = pl + 1, s1 = pr - 2;
  const int nscan = s1 - s0 + 1;
  const int W = (nscan + 63) >> 6;
  // chunks of 2^cs words (cs >= 2: one 4-word group of the flag pass), at most MAXCH of them
  int cs = 2;
  while (((W + (1 << cs) - 1) >> cs) > MAXCH) cs++;
  const int nch = (W + (1 << cs) - 1) >> cs;
  int nchp = 1;
  while (nchp < nch) nchp <<= 1;
  PROF_DECL;

  // ---- A: median of three (on keys), pivot parked at pr-1 (quicksort.cpp, head of the partition loop)
  if (tid == 0) {
    const int pm = pl + ((pr - pl) >> 1);
    uint32_t el = A.get(pl), em = A.get(pm), er = A.get(pr), t;
    const uint32_t e2 = A.get(pr - 1);
    uint32_t kl = norm_key(eraw(el), fmx), km = norm_key(eraw(em), fmx), kr = norm_key(eraw(er), fmx);
    if (km < kl) { t = em; em = el; el = t; t = km; km = kl; kl = t; }
    if (kr < km) { t = er; er = em; em = t; t = kr; kr = km; km = t; }
    if (km < kl) { t = em; em = el; el = t; }
    A.set(pl, el); A.set(pr, er); A.set(pm, e2); A.set(pr - 1, em);
    sh->xp = eraw(em);
  }
  gsync<WAVE>();
  PROF(WAVE ? 11 : 1);
  uint32_t tlo, thi;                    // key(raw) >= key(pivot) <=> raw >= tlo;  key(raw) <= key(pivot) <=> raw <= thi
  key_class(uni(sh->xp), fmx, tlo, thi);
  tlo = uni(tlo); thi = uni(thi);

  // ---- B: stopper masks of the scan region + running counts inside each chunk; four words (256 positions) per step,
  //         loads unconditional (clamped address), validity as a wave-uniform mask, one lane stores the group
  // (round 5: the per-word popcounts and running sums moved from the scalar unit -- ONE per CU, shared by the 16 waves of two rows, and
  //  the busiest unit of this kernel -- to the vector lanes: lane u of the wave takes word u of the group, counts its two masks, a
  //  4-lane DPP prefix gives the running counts, and lanes 0-3 store their own words: N = 16 000 1.24 -> 1.15 ms.  Requesting the next group's
  //  entries ahead of the current group's chain measured slower, 1.18 ms)
  for (int ch = wav; ch < nch; ch += nwav) {
    uint32_t run = 0;                                    // running stopper counts of the chunk so far: L | R << 16 (a chunk holds <= 4096 positions)
    const int wb = ch << cs, we = min(W, wb + (1 << cs));
    for (int w = wb; w < we; w += 4) {
      uint32_t e[4];
#pragma unroll
      for (int u = 0; u < 4; u++) e[u] = A.get(min(s0 + ((w + u) << 6) + lane, s1));
      uint64_t lm[4], rm[4];
      // all four words inside the scan region (every group but the last of a partition): no validity masks
      if (((w + 4) << 6) <= nscan) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t x = eraw(e[u]);
          lm[u] = __builtin_amdgcn_uicmp(x, tlo, 35);                     // ICMP_UGE
          rm[u] = __builtin_amdgcn_uicmp(x, thi, 37);                     // ICMP_ULE
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int rem = nscan - ((w + u) << 6);                         // scan positions left from this word on (uniform)
          const uint64_t vm = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
          const uint32_t x = eraw(e[u]);
          lm[u] = __builtin_amdgcn_uicmp(x, tlo, 35) & vm;                // ICMP_UGE
          rm[u] = __builtin_amdgcn_uicmp(x, thi, 37) & vm;                // ICMP_ULE
        }
      }
      const uint64_t myL = lane == 0 ? lm[0] : (lane == 1 ? lm[1] : (lane == 2 ? lm[2] : lm[3]));
      const uint64_t myR = lane == 0 ? rm[0] : (lane == 1 ? rm[1] : (lane == 2 ? rm[2] : rm[3]));
      const uint32_t c = (uint32_t)__popcll(myL) | ((uint32_t)__popcll(myR) << 16);
      const uint32_t t = c + dpp_mov<0x111, 0xf, 0xf, true>(c);          // + lane - 1 (row_shr:1, zero fill)
      const uint32_t inc = t + dpp_mov<0x112, 0xf, 0xf, true>(t) + run;  // + lanes - 2, - 3: inclusive counts of words w .. w + lane
      if (lane < 4) { mk.L[w + lane] = myL; mk.R[w + lane] = myR; mk.P[w + lane] = inc; }     // the arrays are padded to a multiple of four words
      run = (uint32_t)__builtin_amdgcn_readlane((int)inc, 3);
    }
    if (lane == 0) { sh->ctotL[ch] = run & 0xffffu; sh->ctotR[ch] = run >> 16; }
  }
  gsync<WAVE>();
  PROF(WAVE ? 12 : 2);

  // ---- C: chunk prefix in registers (every wave scans the <= 64 chunk totals itself: no barrier, no shared result)
  const uint32_t cL = lane < nch ? sh->ctotL[lane] : 0u, cR = lane < nch ? sh->ctotR[lane] : 0u;
  const uint32_t inL = wave_incl_scan(cL), inR = wave_incl_scan(cR);
  const uint32_t exL = inL - cL, exR = inR - cR;
  const uint32_t totL = (uint32_t)__builtin_amdgcn_readlane((int)inL, 63), totR = (uint32_t)__builtin_amdgcn_readlane((int)inR, 63);
  // crossing word = number of words at whose end g >= f still holds (monotone), counted 64 words at a time
  int wstar = 0;
  for (int wb = 0; wb < W; wb += 64) {   // (W, cs, nch are scalars: pl and pr were made uniform above)
    const int w = wb + lane;
    const bool valid = w < W;
    const int wc = valid ? w : W - 1;
    const uint32_t oL = (uint32_t)__shfl((int)exL, wc >> cs), oR = (uint32_t)__shfl((int)exR, wc >> cs);
    const uint32_t pw = mk.P[wc];
    const bool g = valid && (totR - ((pw >> 16) + oR)) >= ((pw & 0xffffu) + oL);
    const int pc = __popcll(__ballot(g));
    wstar += pc;
    if (pc != 64) break;                 // monotone: the first false (or the end of the words) ends the count
  }
  PROF(WAVE ? 13 : 3);

  // ---- D: number of swaps m and the pivot position p* + 1 (every wave computes them redundantly)
  uint32_t m;
  int pstar;
  if (wstar >= W) {
    m = totL; pstar = s1;
  } else {
    const uint64_t lm = uni(mk.L[wstar]), rm = uni(mk.R[wstar]);
    const int ch = wstar >> cs;
    const uint32_t pw = uni(mk.P[wstar]);
    const uint32_t pL = (pw & 0xffffu) + uni((uint32_t)__shfl((int)exL, ch)), pR = (pw >> 16) + uni((uint32_t)__shfl((int)exR, ch));
    const uint32_t cumL = pL - (uint32_t)__popcll(lm);       // L-stoppers in words < wstar
    const uint32_t cumR = totR - pR;                         // R-stoppers in words > wstar
    const uint64_t le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const uint32_t f = cumL + (uint32_t)__popcll(lm & le);
    const uint32_t g = cumR + (uint32_t)__popcll(rm & ~le);
    const bool valid = (wstar << 6) + lane < nscan;
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
  m = uni(m);
  const int pi = uni(pstar) + 1;
  PROF(WAVE ? 14 : 4);

  // ---- E: swap the k-th L-stopper from the left with the k-th R-stopper from the right, k <= m
  {
    const uint32_t c = (m + nthr - 1) / nthr;
    const uint32_t k0 = (uint32_t)tid * c + 1u;
    const uint32_t k1 = min(m, k0 + c - 1u);
    const bool work = m != 0 && k0 <= k1;
    // the chunk search shuffles across the wave: every lane takes part (idle lanes look up rank 1 of an existing stopper)
    if (__ballot(work)) {
      const uint32_t rl = work ? k0 : 1u, rr = work ? (totR - k0 + 1u) : 1u;      // the R stopper counted from the left
      uint32_t offl, offr;
      int wl = word_of_rank<false>(mk.P, exL, W, cs, nch, nchp, rl, offl);
      int wr = word_of_rank<true>(mk.P, exR, W, cs, nch, nchp, rr, offr);
      if (work) {
        uint64_t ml = mk.L[wl], mr = mk.R[wr];
        {
          const uint32_t before = offl + (mk.P[wl] & 0xffffu) - (uint32_t)__popcll(ml);
          const int b = nth_set_bit(ml, (int)(rl - before));
          ml &= ~((1ull << b) - 1ull);                               // keep bits >= b
        }
        {
          const uint32_t before = offr + (mk.P[wr] >> 16) - (uint32_t)__popcll(mr);
          const int b = nth_set_bit(mr, (int)(rr - before));
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
  }
  gsync<WAVE>();
  PROF(WAVE ? 15 : 5);

  // ---- F: pivot into place (visible to thread 0 / wave 0, which own every step that follows before the next barrier)
  if (tid == 0) {
    const uint32_t e1 = A.get(pi), e2 = A.get(pr - 1);
    A.set(pi, e2); A.set(pr - 1, e1);
  }
  PROF(6);
  return pi;
}

// heapsort.cpp aheapsort_ on A[lo .. lo+n) (one thread; only reached past the depth budget)
template <class Arena>
__device__ __forceinline__ void heapsort(const Arena& A, float fmx, int lo, int n) {
  const int base = lo - 1;   // 1-based heap
  auto key = [&](int i) { return norm_key(eraw(A.get(base + i)), fmx); };
  int i, j, l;
  uint32_t tmp, ktmp;
  for (l = n >> 1; l > 0; --l) {
    tmp = A.get(base + l); ktmp = norm_key(eraw(tmp), fmx);
    for (i = l, j = l << 1; j <= n;) {
      if (j < n && key(j) < key(j + 1)) j += 1;
      if (ktmp < key(j)) { A.set(base + i, A.get(base + j)); i = j; j += j; } else break;
    }
    A.set(base + i, tmp);
  }
  for (; n > 1;) {
    tmp = A.get(base + n); ktmp = norm_key(eraw(tmp), fmx); A.set(base + n, A.get(base + 1)); n -= 1;
    for (i = 1, j = 2; j <= n;) {
      if (j < n && key(j) < key(j + 1)) j++;
      if (ktmp < key(j)) { A.set(base + i, A.get(base + j)); i = j; j += j; } else break;
    }
    A.set(base + i, tmp);
  }
}

// insertion sort of A[pl..pr] (<= 16 entries) == stable rank sort, one wave
template <class Arena>
__device__ __forceinline__ void insertion(const Arena& A, float fmx, int pl, int pr) {
  gsync<true>();
  const int lane = lane_id(), n = pr - pl + 1;
  const uint32_t e = lane < n ? A.get(pl + lane) : 0u;
  const uint32_t k = norm_key(eraw(e), fmx);
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
// tailn > 0: split mode -- ranges of at most tailn entries are recorded in sh->tail (sh->ntail) instead of being finished here;
// tailn == 0: everything is finished in this kernel (ranges of at most WAVE_N entries by wave 0 alone, the round-2 behaviour).
template <int NT, class Arena>
__device__ __forceinline__ void sort_prefix(const Arena& A, Ctl* sh, const Masks& mk, float fmx, int N, int K, int tailn, int cd0 PROF_ARG) {
  const int tid = (int)threadIdx.x, lane = lane_id(), wav = tid >> 6;
  const int wave_n = tailn > 0 ? tailn : WAVE_N;
  int sp = 0, ntail = 0;
  int pl = 0, pr = N - 1, cd = cd0;      // cd0 = 2 * floor(log2 N) for a whole row (numpy's depth budget)
  bool have = true;
  for (;;) {
    if (!have) {
      if (sp == 0) break;
      __syncthreads();
      --sp;
      pl = uni(sh->stack[3 * sp]); pr = uni(sh->stack[3 * sp + 1]); cd = uni(sh->stack[3 * sp + 2]);
      if (cd < 0 && !(tailn > 0 && pr - pl + 1 <= tailn)) {   // popped past the depth budget: heapsort the whole range (short ones: in the tail kernel)
        if (tid == 0) heapsort(A, fmx, pl, pr - pl + 1);
        continue;
      }
    }
    have = false;
    bool needed = true;
    if (tailn > 0) {
      const bool spent = cd < 0;          // only a short popped range reaches this point with a spent budget
      if (!spent) {
        while (pr - pl > SMALL && pr - pl + 1 > tailn) {
          const int pi = partition<false, NT>(A, sh, mk, fmx, pl, pr PROF_PASS);
          --cd;
          const Split s = split_ranges(pl, pr, pi);
          if (s.ql < K && s.ql <= s.qr) {
            if (tid == 0) { sh->stack[3 * sp] = s.ql; sh->stack[3 * sp + 1] = s.qr; sh->stack[3 * sp + 2] = cd; }
            ++sp;
          }
          if (s.cl < K && s.cl <= s.cr) { pl = s.cl; pr = s.cr; }
          else { needed = false; break; }
        }
      }
      if (needed && pr > pl) {            // at most tailn entries left (or a short range past its depth budget): the tail kernel's
        // numpy tests the depth budget only when a range is POPPED (top of aquicksort's outer loop), never while it keeps
        // partitioning the child it continues with: a range that passed the test here carries NOCHECK, a popped short range
        // whose budget is spent (cd < 0) carries its raw cd and is heapsorted by the tail kernel
        if (tid == 0) { sh->tail[3 * ntail] = pl; sh->tail[3 * ntail + 1] = pr; sh->tail[3 * ntail + 2] = spent ? cd : cd + NOCHECK; }
        ++ntail;
      }
      continue;
    }
    while (pr - pl > SMALL && pr - pl + 1 > wave_n) {
      const int pi = partition<false, NT>(A, sh, mk, fmx, pl, pr PROF_PASS);
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
          const int pi = partition<true, NT>(A, sh, mk, fmx, pl, pr PROF_PASS);
          --cd;
          const Split s = split_ranges(pl, pr, pi);
          if (s.ql < K && s.ql <= s.qr) {
            if (lane == 0) { sh->stack[3 * wsp] = s.ql; sh->stack[3 * wsp + 1] = s.qr; sh->stack[3 * wsp + 2] = cd; }
            ++wsp;
          }
          if (s.cl < K && s.cl <= s.cr) { pl = s.cl; pr = s.cr; }
          else { needed = false; break; }
        }
        { PROF_DECL; if (needed && pr > pl) insertion(A, fmx, pl, pr); PROF(8); }
        if (lane == 0) sh->sp = wsp;
      }
      __syncthreads();
      sp = uni(sh->sp);
    }
  }
  if (tid == 0) sh->ntail = ntail;
  __syncthreads();
}

// the tail kernel's loop: one wave finishes the ranges on sh->stack (sp of them) -- the wave-level half of sort_prefix
template <int NT, class Arena>
__device__ __forceinline__ void sort_tail(const Arena& A, Ctl* sh, const Masks& mk, float fmx, int K, int sp PROF_ARG) {
  const int lane = lane_id();
  while (sp > 0) {
    gsync<true>();
    --sp;
    int pl = uni(sh->stack[3 * sp]), pr = uni(sh->stack[3 * sp + 1]), cd = uni(sh->stack[3 * sp + 2]);
    if (cd >= NOCHECK / 2) cd -= NOCHECK;   // handed over in the middle of its partition loop: no depth test (see sort_prefix)
    else if (cd < 0) {                    // popped past the depth budget: heapsort the whole range
      if (lane == 0) heapsort(A, fmx, pl, pr - pl + 1);
      continue;
    }
    bool needed = true;
    while (pr - pl > SMALL) {
      const int pi = partition<true, NT>(A, sh, mk, fmx, pl, pr PROF_PASS);
      --cd;
      const Split s = split_ranges(pl, pr, pi);
      if (s.ql < K && s.ql <= s.qr) {
        if (lane == 0) { sh->stack[3 * sp] = s.ql; sh->stack[3 * sp + 1] = s.qr; sh->stack[3 * sp + 2] = cd; }
        ++sp;
      }
      if (s.cl < K && s.cl <= s.cr) { pl = s.cl; pr = s.cr; }
      else { needed = false; break; }
    }
    if (needed && pr > pl) insertion(A, fmx, pl, pr);
  }
  gsync<true>();
}

// LDS: Ctl | L[wcap] R[wcap] (u64) | P[wcap] (u32) | entries.  Entry of column j lives at word `first + j`, where
// `first` = misalignment of the row start in D (in halves, 0..7): every 16-byte load of D then maps to two aligned 16-byte
// stores of entries (conflict-free), whatever the row's alignment.
template <bool LDS, int NT>
__global__ __launch_bounds__(NT) void topk_introsort_kernel(const hbits* __restrict__ D, const unsigned* __restrict__ rowmax, int N, int nrows,
                                                            int K, int wcap, uint32_t* __restrict__ arena, size_t arena_stride,
                                                            int32_t* __restrict__ rank, int tailn, uint32_t* __restrict__ tails, size_t tail_stride,
                                                            const int* __restrict__ only) {
  extern __shared__ __align__(16) unsigned char smem[];
  Ctl* sh = reinterpret_cast<Ctl*>(smem);
  Masks mk;
  mk.L = reinterpret_cast<uint64_t*>(smem + ((sizeof(Ctl) + 15) & ~(size_t)15));
  mk.R = mk.L + wcap;
  mk.P = reinterpret_cast<uint32_t*>(mk.R + wcap);
  uint32_t* ent = mk.P + wcap;            // wcap is a multiple of 4: 16-byte aligned
  const int tid = (int)threadIdx.x;
  for (int row = (int)blockIdx.x; row < nrows; row += (int)gridDim.x) {
    if (only && only[row] == 0) continue;   // (uniform) fallback launch behind the streamed kernel: only the rows it flagged
#ifdef SSG_INTRO_PROF
    ProfAcc pacc_;
#pragma unroll
    for (int i_ = 0; i_ < 16; i_++) pacc_.a[i_] = 0;
#endif
    PROF_DECL;
    const float fmx = h2f((hbits)rowmax[row]);
    // ---- the row as packed (raw half, column) entries
    const int64_t total = (int64_t)nrows * N;
    const int64_t base = (int64_t)row * N;
    const int64_t al = base & ~(int64_t)7;
    const int first = (int)(base - al);
    const int nch = (first + N + 7) >> 3;
    uint32_t* dst = LDS ? ent : arena + (size_t)blockIdx.x * arena_stride;
    // (round 5: the row goes through a bounds-checked buffer resource that starts at its 16-byte-aligned base and ends with the matrix:
    //  reads past the end return 0, so the loads are unconditional straight-line code.  With `if (c < nch) { if (off + 8 <= total) load
    //  else tail }` around them hipcc waited for each load where its branch joined -- the "loads in flight" were one at a time.)
    // (whole dwords: the range check is per dword, and the last half of an odd-sized matrix shares its dword with 2 bytes behind the end)
    const int64_t row_bytes_ = ((total - al) * 2 + 3) & ~(int64_t)3;
    const __amdgpu_buffer_rsrc_t rrow_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<hbits*>(D + al), 0,
                                                                        (int)(row_bytes_ > 0x7ffffff0ll ? 0x7ffffff0ll : row_bytes_), 0x00020000);
    constexpr int LU = 4;                              // 16-byte loads in flight per thread
    for (int c0 = 0; c0 < nch; c0 += LU * NT) {
      uint4 x[LU];
#pragma unroll
      for (int u = 0; u < LU; u++) {
        typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
        const v4u_ t_ = __builtin_amdgcn_raw_buffer_load_b128(rrow_, min(c0 + u * NT + tid, nch - 1) * 16, 0, 0);
        x[u] = make_uint4(t_[0], t_[1], t_[2], t_[3]);
      }
#pragma unroll
      for (int u = 0; u < LU; u++) {
        const int c = c0 + u * NT + tid;
        if (c < nch) {
          const unsigned w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
          const uint32_t j0 = (uint32_t)(c * 8 - first);    // columns outside [0, N) land in the padding words around the row
          uint32_t ev[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const uint32_t r = (w[e >> 1] >> ((e & 1) * 16)) & 0x7fffu;   // distances are non-negative: bit 15 is never set
            ev[e] = (r << IDX_BITS) | ((j0 + (uint32_t)e) & IDX_MASK);
          }
          uint4* q = reinterpret_cast<uint4*>(dst + (size_t)c * 8);
          q[0] = make_uint4(ev[0], ev[1], ev[2], ev[3]);
          q[1] = make_uint4(ev[4], ev[5], ev[6], ev[7]);
        }
      }
    }
    __syncthreads();
    PROF(0);
    auto finish = [&](const auto& A) {
      if (tailn == 0) {
        if (tid < K) rank[(int64_t)row * K + tid] = (int32_t)(A.get(tid) & IDX_MASK);
        return;
      }
      // split mode: hand the short ranges and the entries they cover (at least the first K) to the tail kernel
      const int nt = sh->ntail;
      int hi = K;
      for (int q = 0; q < nt; q++) hi = max(hi, sh->tail[3 * q + 1] + 1);
      uint32_t* trow = tails + (size_t)row * tail_stride;
      if (tid == 0) { trow[0] = (uint32_t)nt; trow[1] = (uint32_t)hi; }
      for (int q = tid; q < 3 * nt; q += NT) trow[4 + q] = (uint32_t)sh->tail[q];
      for (int j = tid; j < hi; j += NT) trow[TAIL_HDR_WORDS + j] = A.get(j);
    };
    if (LDS) {
      LdsArena A{ent + first};
      sort_prefix<NT>(A, sh, mk, fmx, N, K, tailn, 2 * (31 - __clz(N)) PROF_PASS);
      PROF(9);
      finish(A);
    } else {
      GlobalArena A{dst + first};
      sort_prefix<NT>(A, sh, mk, fmx, N, K, tailn, 2 * (31 - __clz(N)) PROF_PASS);
      finish(A);
    }
    __syncthreads();
#ifdef SSG_INTRO_PROF
    if (tid == 0) {
#pragma unroll
      for (int i_ = 0; i_ < 16; i_++) if (pacc_.a[i_]) atomicAdd(&g_prof[i_], pacc_.a[i_]);
    }
#endif
  }
}

// One wave per row: the ranges the workgroup kernel handed over (at most tailn entries each, all inside [0, K + tailn)) are
// partitioned / insertion-sorted / heapsorted exactly as wave 0 of the unsplit kernel would, on an LDS copy of the row's head.
__global__ __launch_bounds__(64) void topk_introsort_tail_kernel(const uint32_t* __restrict__ tails, size_t tail_stride, const unsigned* __restrict__ rowmax,
                                                                 int nrows, int K, int wcap, int32_t* __restrict__ rank) {
  extern __shared__ __align__(16) unsigned char smem[];
  Ctl* sh = reinterpret_cast<Ctl*>(smem);
  Masks mk;
  mk.L = reinterpret_cast<uint64_t*>(smem + ((sizeof(Ctl) + 15) & ~(size_t)15));
  mk.R = mk.L + wcap;
  mk.P = reinterpret_cast<uint32_t*>(mk.R + wcap);
  uint32_t* ent = mk.P + wcap;
  const int lane = lane_id();
  for (int row = (int)blockIdx.x; row < nrows; row += (int)gridDim.x) {
#ifdef SSG_INTRO_PROF
    ProfAcc pacc_;
#pragma unroll
    for (int i_ = 0; i_ < 16; i_++) pacc_.a[i_] = 0;
#endif
    const uint32_t* trow = tails + (size_t)row * tail_stride;
    const int nt = uni((int)trow[0]), hi = uni((int)trow[1]);
    const float fmx = h2f((hbits)rowmax[row]);
    for (int j = lane * 4; j < hi; j += 256) *reinterpret_cast<uint4*>(ent + j) = *reinterpret_cast<const uint4*>(trow + TAIL_HDR_WORDS + j);
    for (int q = lane; q < 3 * nt; q += 64) sh->stack[q] = (int)trow[4 + q];
    gsync<true>();
    LdsArena A{ent};
    sort_tail<64>(A, sh, mk, fmx, K, nt PROF_PASS);
    if (lane < K) rank[(int64_t)row * K + lane] = (int32_t)(A.get(lane) & IDX_MASK);
    gsync<true>();
#ifdef SSG_INTRO_PROF
    if (lane == 0) {
#pragma unroll
      for (int i_ = 0; i_ < 16; i_++) if (pacc_.a[i_]) atomicAdd(&g_prof[i_], pacc_.a[i_]);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: rows that do not fit in LDS (N > ~36 k) -- the first levels of the walk as OUT-OF-PLACE, streamed partitions.
//
// The in-place kernel above keeps such a row in a global arena and performs every swap of every level there: N/4 random 4-byte
// read-modify-writes of a 512 KB row at level 1 alone (configs[4], N = 128 000: 350 ms, 0.012 of the HBM rate of the bytes it reads).
// On the main path of the walk only the LEFT child of a partition is needed (every range that intersects columns [0, K) starts at 0
// until a pivot lands inside [0, K)), and after the m swaps the left child is
//     left[0] = the smallest median candidate;  left[p] = A[p] for p <= p* that are not L-stoppers;
//     left[p] = A[R_k] for the k-th L-stopper from the left (R_k = the k-th R-stopper from the right),
// where A is the source as the median-of-3 step leaves it (positions 0 and pm overridden).  So a level READS its source in order (the
// row of D itself at level 1: 2 bytes per column, entries are built on the fly -- the row is never materialised at full length),
// builds the stopper masks as bytes (8 positions per lane and 16-byte load), and WRITES the left child once: into LDS when it fits
// (then the LDS code above takes over), else into one of two global buffers of the workgroup.  The pairing goes through a staging
// buffer in LDS, RBUF ranks at a time: pass X collects the R-stoppers of ranks [k0, k1] in rank order from the mask words at the right
// end, pass Y hands them to the L-stoppers of the same ranks at the left end -- both passes touch whole mask words with one lane per
// bit, every global access is coalesced, and no rank -> position search runs per element.
// tools/introsort_model.py (stream_partition_left) states the same computation word by word and is checked against the in-place model.
// A pivot that lands inside [0, K) (it is an output itself, and the right child may matter too) is not handled here: the row is flagged
// and redone by the in-place kernel.
#ifndef SSG_STREAM_RBUF
#define SSG_STREAM_RBUF 8192
#endif
#ifndef SSG_STREAM_SU
#define SSG_STREAM_SU 4
#endif
#ifndef SSG_STREAM_SX
#define SSG_STREAM_SX 8
#endif
constexpr int RBUF = SSG_STREAM_RBUF;  // ranks per staging pass (32 KB of LDS; 4096: +4 % time at N = 128 000, 8192 leaves 22 k entries of LDS for the row)
constexpr int SU = SSG_STREAM_SU;      // 8-entry groups a thread keeps in flight in passes B1 / E1
constexpr int SX = SSG_STREAM_SX;      // mask words a wave keeps in flight in passes X / Y

// What a streamed level reads; position p lives at index q = p + qs.  Two kinds, a compile-time switch (a run-time branch in front of
// every load made hipcc wait for each load before it issued the next: the loads of a batch must be straight-line code):
//   DROW: the row of D itself, halves, through a buffer resource that starts at the row's 16-byte-aligned base and ends with the
//         matrix (reads past the end return 0: the last group of the last row needs no special case); entries are built on the fly;
//   else: materialised entries arr[q] in one of the workgroup's global buffers (always read in bounds).
struct Raw8 { uint4 a, b; };
template <bool DROW> struct Src {
  const uint32_t* arr; __amdgpu_buffer_rsrc_t rs; int qs;
  __device__ __forceinline__ uint32_t get(int p) const {
    const int q = p + qs;
    if constexpr (DROW) {
      const uint32_t raw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, q * 2, 0, 0) & 0x7fffu;
      return (raw << IDX_BITS) | ((uint32_t)p & IDX_MASK);
    } else return arr[q];
  }
  __device__ __forceinline__ void load8(int c, Raw8& r) const {
    if constexpr (DROW) {
      typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
      const v4u_ x = __builtin_amdgcn_raw_buffer_load_b128(rs, c * 16, 0, 0);
      r.a = make_uint4(x[0], x[1], x[2], x[3]);
    } else { r.a = *reinterpret_cast<const uint4*>(arr + (size_t)c * 8); r.b = *reinterpret_cast<const uint4*>(arr + (size_t)c * 8 + 4); }
  }
  __device__ __forceinline__ void decode8(int c, const Raw8& r, uint32_t (&e)[8]) const {
    if constexpr (DROW) {
      const unsigned w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
      for (int i = 0; i < 8; i++) e[i] = (((w[i >> 1] >> ((i & 1) * 16)) & 0x7fffu) << IDX_BITS) | ((uint32_t)(c * 8 + i - qs) & IDX_MASK);
    } else { e[0] = r.a.x; e[1] = r.a.y; e[2] = r.a.z; e[3] = r.a.w; e[4] = r.b.x; e[5] = r.b.y; e[6] = r.b.z; e[7] = r.b.w; }
  }
  // stopper bits of the 8 positions of a group, no per-position range / override checks (the caller redoes the few groups that need them):
  // raw >= tlo <=> bit 15 of (raw | 0x8000) - tlo;  raw <= thi <=> bit 15 of (thi | 0x8000) - raw  (raw, tlo, thi < 0x8000) -- two halves per
  // dword for the row of D (no borrow crosses the halves: the low half's difference is never negative)
  __device__ __forceinline__ void stopper_bits8(const Raw8& r, uint32_t tlo, uint32_t thi, unsigned& lb, unsigned& rb) const {
    lb = 0; rb = 0;
    if constexpr (DROW) {
      const unsigned w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
      const uint32_t tl2 = tlo * 0x10001u, th2 = (thi | 0x8000u) * 0x10001u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t x = w[j] & 0x7fff7fffu;
        const uint32_t tL = (x | 0x80008000u) - tl2, tR = th2 - x;
        lb |= (((tL >> 15) & 1u) | ((tL >> 30) & 2u)) << (2 * j);
        rb |= (((tR >> 15) & 1u) | ((tR >> 30) & 2u)) << (2 * j);
      }
    } else {
      const uint32_t e[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
      const uint32_t th = thi | 0x8000u;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t x = e[i] >> IDX_BITS;
        lb |= ((((x | 0x8000u) - tlo) >> 15) & 1u) << i;
        rb |= (((th - x) >> 15) & 1u) << i;
      }
    }
  }
};

// word holding the r-th (1-based, from the left) stopper, two ballots instead of word_of_rank's dependent binary searches: the chunk is
// the last one whose exclusive offset (lane c of `ex`) is below r, the word inside it the first whose running count reaches r - offset
template <bool RIGHT>
__device__ __forceinline__ int word_of_rank_ballot(const uint32_t* P, uint32_t ex, int W, int cs, int nch, uint32_t r) {
  const int lane = lane_id();
  const int ch = __popcll(__ballot(lane < nch && ex < r)) - 1;            // (ex[0] = 0 < r)
  const uint32_t rr = r - (uint32_t)__builtin_amdgcn_readlane((int)ex, ch);
  const int w = (ch << cs) + lane;
  const bool valid = lane < (1 << cs) && w < W;
  const uint32_t pw = valid ? P[w] : 0u;
  const uint32_t c = RIGHT ? (pw >> 16) : (pw & 0xffffu);
  return (ch << cs) + __popcll(__ballot(valid && c < rr));
}

// passes E1 / X / Y of a streamed partition: the left child [0, pi) into tgt (index q = p + qs; LDS or global memory).
// (PMC at N = 128 000: 160 GB fetched + 102 GB written per launch for 2 N^2 = 32.8 GB of D: every level reads its source twice -- masks,
// then the copy -- plus the sparse gather, and pass Y's 4-byte stores hit half of the lines pass E1 has just written.  A single pass that
// writes every position of the child once -- the L-stoppers' staged partners and the other positions' own entries, 64 consecutive entries
// per wave store -- was built and measured: 76 -> 81 ms; per-lane 2- / 4-byte source reads for ALL positions cost more than the 16-byte
// vector copy + the scattered stores they replace.)
template <int NT, bool DROW, class Tgt>
__device__ __forceinline__ void stream_emit(const Src<DROW>& s, Tgt tgt, uint32_t* rbuf, const Masks& mk, int n, int W, int cs, int nch, uint32_t exL, uint32_t exR,
                                            uint32_t totR, uint32_t m, int pstar, int pm, uint32_t el, uint32_t e2 PROF_ARG) {
  constexpr int NW = NT / 64;
  const int tid = (int)threadIdx.x, lane = lane_id(), wav = __builtin_amdgcn_readfirstlane(tid >> 6), qs = s.qs;
  PROF_DECL;
  // E1: every position of the left child from the source (coalesced); the L-stopper positions are overwritten by pass Y.  The two
  // positions the median-of-3 step changed (0 and pm) are patched by the thread that stored their group, right behind its store.
  const int clast = (pstar + qs) >> 3;
  const int c_pm = pm <= pstar ? ((pm + qs) >> 3) : -1;
  for (int c0 = tid; c0 <= clast; c0 += SU * NT) {
    Raw8 raw[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) s.load8(min(c0 + u * NT, clast), raw[u]);          // (clamped, unconditional: straight-line loads)
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int c = c0 + u * NT;
      if (c <= clast) {
        uint32_t e[8];
        s.decode8(c, raw[u], e);
        *reinterpret_cast<uint4*>(&tgt[c * 8]) = make_uint4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<uint4*>(&tgt[c * 8 + 4]) = make_uint4(e[4], e[5], e[6], e[7]);
        if (c == c_pm) tgt[pm + qs] = e2;
        if (c == 0) tgt[qs] = el;
      }
    }
  }
  __syncthreads();
  PROF(4);
  const uint64_t le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  for (uint32_t k0 = 1; k0 <= m; k0 += RBUF) {          // (m is workgroup-uniform: every thread runs the same barriers)
    const uint32_t k1 = min(m, k0 + RBUF - 1u);
    // X: the R-stoppers of ranks k0..k1 from the right = left ranks totR-k1+1 .. totR-k0+1, in rank order into rbuf.  SX mask words per
    // trip and wave: mask words first (LDS broadcasts), then every lane's global read (unconditional, clamped), then the LDS stores
    const int wa = uni(word_of_rank_ballot<true>(mk.P, exR, W, cs, nch, totR - k1 + 1u));
    const int wb = uni(word_of_rank_ballot<true>(mk.P, exR, W, cs, nch, totR - k0 + 1u));
    PROF(8);
    for (int w0 = wa + wav; w0 <= wb; w0 += SX * NW) {
      uint64_t rm[SX]; uint32_t pw[SX];
#pragma unroll
      for (int u = 0; u < SX; u++) { const int w = min(w0 + u * NW, wb); rm[u] = mk.R[w]; pw[u] = mk.P[w]; }
      uint32_t val[SX];
#pragma unroll
      for (int u = 0; u < SX; u++) val[u] = s.get(min(max((min(w0 + u * NW, wb) << 6) + lane - qs, 0), n - 1));
#pragma unroll
      for (int u = 0; u < SX; u++) {
        const int w = w0 + u * NW;
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)exR, min(w, wb) >> cs) + (pw[u] >> 16) - (uint32_t)__popcll(rm[u]);
        const uint32_t kk = totR - (before + (uint32_t)__popcll(rm[u] & le)) + 1u;
        if (w <= wb && ((rm[u] >> lane) & 1ull) && kk >= k0 && kk <= k1) rbuf[kk - k0] = ((w << 6) + lane - qs == pm) ? e2 : val[u];
      }
    }
    PROF(9);
    __syncthreads();
    PROF(5);
    // Y: the L-stoppers of ranks k0..k1 take them
    const int wc = uni(word_of_rank_ballot<false>(mk.P, exL, W, cs, nch, k0));
    const int wd = uni(word_of_rank_ballot<false>(mk.P, exL, W, cs, nch, k1));
    PROF(10);
    for (int w0 = wc + wav; w0 <= wd; w0 += SX * NW) {
      uint64_t lm[SX]; uint32_t pw[SX];
#pragma unroll
      for (int u = 0; u < SX; u++) { const int w = min(w0 + u * NW, wd); lm[u] = mk.L[w]; pw[u] = mk.P[w]; }
      uint32_t val[SX]; uint32_t kk[SX];
#pragma unroll
      for (int u = 0; u < SX; u++) {
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)exL, min(w0 + u * NW, wd) >> cs) + (pw[u] & 0xffffu) - (uint32_t)__popcll(lm[u]);
        kk[u] = before + (uint32_t)__popcll(lm[u] & le);
        val[u] = rbuf[min(max(kk[u], k0), k1) - k0];                 // (unconditional, clamped: straight-line LDS reads)
      }
#pragma unroll
      for (int u = 0; u < SX; u++) {
        const int w = w0 + u * NW;
        if (w <= wd && ((lm[u] >> lane) & 1ull) && kk[u] >= k0 && kk[u] <= k1) tgt[(w << 6) + lane] = val[u];
      }
    }
    PROF(11);
    __syncthreads();
    PROF(6);
  }
}

// One streamed partition of the n entries of `s` (n > SMALL + 1).  Returns the pivot position pi = the length of the left child, which is
// written to `lds` (index q; *to_lds = true) when pi + qs + 8 <= lds_cap entries, else to `glob`.
template <int NT, bool DROW>
__device__ __forceinline__ int stream_partition(const Src<DROW>& s, int n, Ctl* sh, const Masks& mk, uint32_t* rbuf, float fmx, uint32_t* lds, int lds_cap,
                                                uint32_t* glob, bool& to_lds PROF_ARG) {
  constexpr int NW = NT / 64;
  const int tid = (int)threadIdx.x, lane = lane_id(), wav = __builtin_amdgcn_readfirstlane(tid >> 6), qs = s.qs;
  PROF_DECL;
  const int pr = n - 1, pm = pr >> 1, s0 = 1, s1 = pr - 2;
  const int W = (n + qs + 63) >> 6;
  int cs = 2;
  while (((W + (1 << cs) - 1) >> cs) > MAXCH) cs++;
  const int nch = (W + (1 << cs) - 1) >> cs;
  // ---- A: median of three on keys (quicksort.cpp), recorded as overrides of the read-only source: A[0] = el, A[pm] = e2 (the old
  //         A[pr-1]); the pivot is parked at pr-1 and A[pr] = er -- both outside everything a left child reads
  if (tid == 0) {
    uint32_t el = s.get(0), em = s.get(pm), er = s.get(pr), t;
    const uint32_t e2 = s.get(pr - 1);
    uint32_t kl = norm_key(eraw(el), fmx), km = norm_key(eraw(em), fmx), kr = norm_key(eraw(er), fmx);
    if (km < kl) { t = em; em = el; el = t; t = km; km = kl; kl = t; }
    if (kr < km) { t = er; er = em; em = t; t = kr; kr = km; km = t; }
    if (km < kl) { t = em; em = el; el = t; }
    sh->ov_el = el; sh->ov_e2 = e2; sh->xp = eraw(em);
  }
  __syncthreads();
  const uint32_t el = uni(sh->ov_el), e2 = uni(sh->ov_e2);
  uint32_t tlo, thi;
  key_class(uni(sh->xp), fmx, tlo, thi);
  tlo = uni(tlo); thi = uni(thi);
  PROF(0);
  // ---- B1: stopper masks, one BYTE per lane and 8 positions (bit q of the mask arrays <-> position q - qs; zero outside the scan region)
  {
    uint8_t* Lb = reinterpret_cast<uint8_t*>(mk.L);
    uint8_t* Rb = reinterpret_cast<uint8_t*>(mk.R);
    const int ngroups = (n + qs + 7) >> 3;
    // groups that hold a position outside the scan region [s0, s1] or the overridden position pm take the per-position path
    const int c_lo = (s0 + qs) >> 3, c_hi = (s1 + qs - 7) >> 3, c_pm = (pm + qs) >> 3;      // groups c_lo < c <= c_hi are wholly inside
    for (int c0 = tid; c0 < W * 8; c0 += SU * NT) {
      Raw8 raw[SU];
#pragma unroll
      for (int u = 0; u < SU; u++) s.load8(min(c0 + u * NT, ngroups - 1), raw[u]);      // (clamped, unconditional)
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int c = c0 + u * NT;
        if (c < W * 8) {
          unsigned lb = 0, rb = 0;
          if (c < ngroups) {
            if (c > c_lo && c <= c_hi && c != c_pm) s.stopper_bits8(raw[u], tlo, thi, lb, rb);
            else {
              uint32_t e[8];
              s.decode8(c, raw[u], e);
#pragma unroll
              for (int i = 0; i < 8; i++) {
                const int p = c * 8 + i - qs;
                const uint32_t x = eraw(p == pm ? e2 : e[i]);
                const bool in = p >= s0 && p <= s1;
                lb |= (in && x >= tlo ? 1u : 0u) << i;
                rb |= (in && x <= thi ? 1u : 0u) << i;
              }
            }
          }
          Lb[c] = (uint8_t)lb; Rb[c] = (uint8_t)rb;
        }
      }
    }
  }
  __syncthreads();
  PROF(1);
  // ---- B2: running stopper counts inside each chunk of 2^cs words (<= 64: one lane per word) and the chunk totals
  for (int ch = wav; ch < nch; ch += NW) {
    const int w = (ch << cs) + lane;
    const bool valid = lane < (1 << cs) && w < W;
    const uint32_t cl = valid ? (uint32_t)__popcll(mk.L[w]) : 0u, cr = valid ? (uint32_t)__popcll(mk.R[w]) : 0u;
    const uint32_t il = wave_incl_scan(cl), ir = wave_incl_scan(cr);
    if (valid) mk.P[w] = il | (ir << 16);
    if (lane == 63) { sh->ctotL[ch] = il; sh->ctotR[ch] = ir; }
  }
  __syncthreads();
  PROF(2);
  // ---- C: chunk prefix in registers, crossing word (as in partition())
  const uint32_t cL = lane < nch ? sh->ctotL[lane] : 0u, cR = lane < nch ? sh->ctotR[lane] : 0u;
  const uint32_t inL = wave_incl_scan(cL), inR = wave_incl_scan(cR);
  const uint32_t exL = inL - cL, exR = inR - cR;
  const uint32_t totL = (uint32_t)__builtin_amdgcn_readlane((int)inL, 63), totR = (uint32_t)__builtin_amdgcn_readlane((int)inR, 63);
  int wstar = 0;
  for (int wb = 0; wb < W; wb += 64) {
    const int w = wb + lane;
    const bool valid = w < W;
    const int wc = valid ? w : W - 1;
    const uint32_t oL = (uint32_t)__shfl((int)exL, wc >> cs), oR = (uint32_t)__shfl((int)exR, wc >> cs);
    const uint32_t pw = mk.P[wc];
    const bool g = valid && (totR - ((pw >> 16) + oR)) >= ((pw & 0xffffu) + oL);
    const int pc = __popcll(__ballot(g));
    wstar += pc;
    if (pc != 64) break;
  }
  // ---- D: number of swaps m and p* (bits outside the scan region are zero in both masks: p* is clamped to [s0 - 1, s1] afterwards)
  uint32_t m;
  int pstar;
  if (wstar >= W) {
    m = totL; pstar = s1;
  } else {
    const uint64_t lm = uni(mk.L[wstar]), rm = uni(mk.R[wstar]);
    const int ch = wstar >> cs;
    const uint32_t pw = uni(mk.P[wstar]);
    const uint32_t pL = (pw & 0xffffu) + uni((uint32_t)__shfl((int)exL, ch)), pR = (pw >> 16) + uni((uint32_t)__shfl((int)exR, ch));
    const uint32_t cumL = pL - (uint32_t)__popcll(lm);
    const uint32_t cumR = totR - pR;
    const uint64_t le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const uint32_t f = cumL + (uint32_t)__popcll(lm & le);
    const uint32_t g = cumR + (uint32_t)__popcll(rm & ~le);
    const uint64_t bal = __ballot(g >= f);
    if (bal == 0) {
      m = cumL; pstar = (wstar << 6) - 1 - qs;
    } else {
      const int b = 63 - __clzll((long long)bal);
      const uint64_t leb = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
      m = cumL + (uint32_t)__popcll(lm & leb);
      pstar = (wstar << 6) + b - qs;
    }
  }
  m = uni(m);
  pstar = uni(pstar);
  pstar = max(s0 - 1, min(pstar, s1));
  const int pi = pstar + 1;
  to_lds = pi + qs + 8 <= lds_cap;
  PROF(3);
  if (to_lds) stream_emit<NT, DROW>(s, lds, rbuf, mk, n, W, cs, nch, exL, exR, totR, m, pstar, pm, el, e2 PROF_PASS);
  else stream_emit<NT, DROW>(s, glob, rbuf, mk, n, W, cs, nch, exL, exR, totR, m, pstar, pm, el, e2 PROF_PASS);
  return pi;
}

// split-mode hand-over of a finished LDS row (or the K ranks when there is no tail kernel)
template <int NT, class Arena>
__device__ __forceinline__ void hand_over(const Arena& A, Ctl* sh, int row, int K, int tailn, uint32_t* __restrict__ tails, size_t tail_stride,
                                          int32_t* __restrict__ rank) {
  const int tid = (int)threadIdx.x;
  if (tailn == 0) {
    if (tid < K) rank[(int64_t)row * K + tid] = (int32_t)(A.get(tid) & IDX_MASK);
    return;
  }
  const int nt = sh->ntail;
  int hi = K;
  for (int q = 0; q < nt; q++) hi = max(hi, sh->tail[3 * q + 1] + 1);
  uint32_t* trow = tails + (size_t)row * tail_stride;
  if (tid == 0) { trow[0] = (uint32_t)nt; trow[1] = (uint32_t)hi; }
  for (int q = tid; q < 3 * nt; q += NT) trow[4 + q] = (uint32_t)sh->tail[q];
  for (int j = tid; j < hi; j += NT) trow[TAIL_HDR_WORDS + j] = A.get(j);
}

// LDS: Ctl | L[wcap] R[wcap] P[wcap] | rbuf[RBUF] | entries[lds_cap].  Persistent workgroups (one per CU: the LDS is full), two global
// buffers of entry_words(N) words each per workgroup for left children that do not fit yet.
template <int NT>
__global__ __launch_bounds__(NT) void topk_introsort_stream_kernel(const hbits* __restrict__ D, const unsigned* __restrict__ rowmax, int N, int nrows, int K,
                                                                   int wcap, uint32_t* __restrict__ bufs, size_t buf_stride, int lds_cap,
                                                                   int32_t* __restrict__ rank, int tailn, uint32_t* __restrict__ tails, size_t tail_stride,
                                                                   int* __restrict__ redo) {
  extern __shared__ __align__(16) unsigned char smem[];
  Ctl* sh = reinterpret_cast<Ctl*>(smem);
  Masks mk;
  mk.L = reinterpret_cast<uint64_t*>(smem + ((sizeof(Ctl) + 15) & ~(size_t)15));
  mk.R = mk.L + wcap;
  mk.P = reinterpret_cast<uint32_t*>(mk.R + wcap);
  uint32_t* rbuf = mk.P + wcap;
  uint32_t* ent = rbuf + RBUF;
  uint32_t* gb[2] = {bufs + (size_t)blockIdx.x * 2 * buf_stride, bufs + ((size_t)blockIdx.x * 2 + 1) * buf_stride};
  const int tid = (int)threadIdx.x;
  const int64_t total = (int64_t)nrows * N;
  for (int row = (int)blockIdx.x; row < nrows; row += (int)gridDim.x) {
    const float fmx = h2f((hbits)rowmax[row]);
    const int64_t base = (int64_t)row * N;
    const int64_t al = base & ~(int64_t)7;
    const int first = (int)(base - al);
    // level 1 reads the row of D through a buffer resource (from the row's aligned base to the end of the matrix, at most 4 GB)
    // (whole dwords: the range check is per dword, and the last half of an odd-sized matrix shares its dword with 2 bytes behind the end)
    const int64_t rs_bytes = ((total - al) * 2 + 3) & ~(int64_t)3;
    Src<true> drow{nullptr, __builtin_amdgcn_make_buffer_rsrc(const_cast<hbits*>(D + al), 0, (int)(rs_bytes > 0x7ffffff0ll ? 0x7ffffff0ll : rs_bytes), 0x00020000), first};
    int n = N, cd = 2 * (31 - __clz(N)), which = 0, level = 0;
    bool in_lds = false, bad = false;
#ifdef SSG_INTRO_PROF
    ProfAcc pacc_;
#pragma unroll
    for (int i_ = 0; i_ < 16; i_++) pacc_.a[i_] = 0;
    unsigned long long prof_lds_ = 0;
#endif
    while (n + first + 8 > lds_cap) {
      bool to_lds;
      int pi;
      if (level == 0) pi = stream_partition<NT, true>(drow, n, sh, mk, rbuf, fmx, ent, lds_cap, gb[which], to_lds PROF_PASS);
      else {
        const Src<false> arr{gb[which ^ 1], drow.rs, first};
        pi = stream_partition<NT, false>(arr, n, sh, mk, rbuf, fmx, ent, lds_cap, gb[which], to_lds PROF_PASS);
      }
      --cd; ++level;
      // pi < K: the pivot itself is an output column (it sits at position pi, which no left child holds), and for pi + 1 < K the right
      // child matters as well: (rare) the row is redone by the in-place kernel
      if (pi < K) { bad = true; break; }
      n = pi;
      if (to_lds) { in_lds = true; break; }
      which ^= 1;
      __syncthreads();                                // the left child in global memory is complete (and visible to every wave) before it is read
    }
    if (bad) {
      if (tid == 0) redo[row] = 1;
      __syncthreads();
      continue;
    }
    if (!in_lds) {                                    // the range fits from the start (a short row forced onto this path): into LDS as it is
      for (int c = tid; c <= ((n - 1 + first) >> 3); c += NT) {
        uint32_t e[8];
        Raw8 r8;
        drow.load8(c, r8);
        drow.decode8(c, r8, e);
        *reinterpret_cast<uint4*>(ent + c * 8) = make_uint4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<uint4*>(ent + c * 8 + 4) = make_uint4(e[4], e[5], e[6], e[7]);
      }
    }
    __syncthreads();
    LdsArena A{ent + first};
#ifdef SSG_INTRO_PROF
    prof_lds_ = clock64();
    ProfAcc pdummy_;        // (the LDS stage's own phase slots are not wanted here: one total)
#pragma unroll
    for (int i_ = 0; i_ < 16; i_++) pdummy_.a[i_] = 0;
    { ProfAcc& pacc2_ = pdummy_; (void)pacc2_; }
    sort_prefix<NT>(A, sh, mk, fmx, n, K, tailn, cd, pdummy_);
#else
    sort_prefix<NT>(A, sh, mk, fmx, n, K, tailn, cd);
#endif
    hand_over<NT>(A, sh, row, K, tailn, tails, tail_stride, rank);
    __syncthreads();
#ifdef SSG_INTRO_PROF
    pacc_.a[7] += clock64() - prof_lds_;
    if (tid == 0) {
#pragma unroll
      for (int i_ = 0; i_ < 16; i_++) if (pacc_.a[i_]) atomicAdd(&g_prof[i_], pacc_.a[i_]);
    }
#endif
  }
}

constexpr size_t LDS_LIMIT = 160 * 1024;
__host__ inline int mask_words(int N) { return (((N + 63) / 64 + 1) + 3) & ~3; }          // padded to the flag pass's 4-word groups
__host__ inline size_t entry_words(int N) { return (size_t)((N + 7 + 7) / 8) * 8; }         // whole 8-entry groups incl. the alignment shift
__host__ inline size_t lds_fixed_bytes(int N) { return ((sizeof(Ctl) + 15) & ~(size_t)15) + (size_t)mask_words(N) * (2 * sizeof(uint64_t) + sizeof(uint32_t)); }
__host__ inline bool fits_lds(int N) { return lds_fixed_bytes(N) + entry_words(N) * 4 <= LDS_LIMIT; }
__host__ inline int arena_blocks(int nrows) { return nrows < 2048 ? nrows : 2048; }

// split mode geometry: hand-over threshold (entries), record stride (32-bit words) and bytes of the hand-over records
__host__ inline int tail_threshold() {
  static int t = -1;
  if (t < 0) { const char* e_ = getenv("SSG_INTRO_TAILN"); t = e_ ? atoi(e_) : 2048; if (t < 0) t = 0; if (t > 8192) t = 8192; }   // 0 = unsplit (round-2 kernel)
  return t;
}
__host__ inline size_t tail_stride_words(int tailn) { return (size_t)TAIL_HDR_WORDS + (size_t)((64 + tailn + 3) & ~3); }
__host__ inline size_t tail_bytes(int nrows) { const int t = tail_threshold(); return t > 0 ? (size_t)nrows * tail_stride_words(t) * 4 : 0; }

// streamed kernel geometry.  Measured (profiles/r05_introsort_stream_timing.txt): two 512-thread workgroups per CU beat one of 1024 while
// the half-LDS entry capacity stays useful (N = 24 000 ... 70 000: -5 ... -18 %), one 1024-thread workgroup with the whole LDS wins when
// the masks alone take most of half the LDS (N = 128 000).  SSG_INTRO_STREAM_CAP / SSG_INTRO_STREAM_NT override (tests, sweeps).
__host__ inline int stream_cap_for(int N, int per_cu) {
  const size_t fixed = lds_fixed_bytes(N) + (size_t)RBUF * 4, budget = LDS_LIMIT / (size_t)per_cu;
  return fixed + 64 < budget ? (int)(((budget - fixed) / 4) & ~(size_t)7) : 0;
}
__host__ inline bool stream_two_per_cu(int N) { return getenv("SSG_INTRO_STREAM_CAP") == nullptr && getenv("SSG_INTRO_STREAM_NT") == nullptr && stream_cap_for(N, 2) >= 4096; }
__host__ inline int stream_nt(int N) {
  const char* e_ = getenv("SSG_INTRO_STREAM_NT");
  const int v = e_ ? atoi(e_) : (stream_two_per_cu(N) ? 512 : 1024);
  return v == 512 ? 512 : 1024;
}
__host__ inline int stream_lds_cap(int N) {
  long cap = stream_cap_for(N, stream_two_per_cu(N) ? 2 : 1);
  const char* e_ = getenv("SSG_INTRO_STREAM_CAP");      // tests: a small capacity makes short rows take several streamed levels (read per call)
  const long lim = e_ ? atol(e_) : 0;
  if (lim > 0 && lim < cap) cap = lim & ~7L;
  return (int)cap;
}
__host__ inline size_t stream_lds_bytes(int N) { return lds_fixed_bytes(N) + (size_t)RBUF * 4 + (size_t)stream_lds_cap(N) * 4; }
// persistent workgroups: as many per CU as the LDS (and the 2048 threads of a CU) hold
__host__ inline int stream_blocks(int N, int nrows) {
  int per_cu = (int)(LDS_LIMIT / stream_lds_bytes(N));
  if (per_cu > 2048 / stream_nt(N)) per_cu = 2048 / stream_nt(N);
  if (per_cu < 1) per_cu = 1;
  return nrows < 256 * per_cu ? nrows : 256 * per_cu;
}
__host__ inline int fallback_blocks(int nrows) { return nrows < 64 ? nrows : 64; }
__host__ inline bool stream_enabled(int N) {
  const char* e_ = getenv("SSG_INTRO_STREAM");          // 0: the in-place arena kernel of rounds 2-4 (read per call)
  return (e_ ? atoi(e_) : 1) != 0 && N >= 4096 && stream_lds_cap(N) >= 2048;
}
// rows-in-LDS kernel: two rows per CU (512 threads each) up to N ~ 18 000; beyond that ONE row per CU -- there the streamed kernel with two
// workgroups per CU is faster (N = 24 000: 3.85 -> 3.17 ms, 30 000: 5.49 -> 4.71, 36 000: 7.42 -> 6.53), so it takes over
__host__ inline bool lds_two_rows(int N) { return 2 * (lds_fixed_bytes(N) + entry_words(N) * 4) <= LDS_LIMIT; }
__host__ inline bool use_stream(int N) {
  if (!fits_lds(N)) return true;                        // (in-place arena kernel when streaming is switched off)
  return !lds_two_rows(N) && stream_enabled(N) && getenv("SSG_INTRO_STREAM_MIDN") == nullptr;     // (SSG_INTRO_STREAM_MIDN set: rows in LDS whenever they fit)
}

template <bool LDS, int NT>
__host__ int launch(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, uint32_t* arena, uint32_t* tails, hipStream_t stream,
                    uint32_t* sbufs = nullptr, int* redo = nullptr) {
  const size_t lds = lds_fixed_bytes(N) + (LDS ? entry_words(N) * 4 : 0);
  const int tailn = tails ? tail_threshold() : 0;             // no hand-over buffer: the unsplit single-launch kernel
  if (!LDS && sbufs) {
    // rows that do not fit in LDS: streamed out-of-place levels first (one workgroup per CU), then the in-place kernel for the rows it flagged
    const int cap = stream_lds_cap(N);
    const size_t slds = stream_lds_bytes(N);
    SSG_HIP(hipMemsetAsync(redo, 0, (size_t)nrows * sizeof(int), stream));
    SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_introsort_stream_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
    hipLaunchKernelGGL((topk_introsort_stream_kernel<NT>), dim3(stream_blocks(N, nrows)), dim3(NT), slds, stream, D, rowmax, N, nrows, K, mask_words(N), sbufs,
                       entry_words(N), cap, rank, tailn, tails, tail_stride_words(tailn), redo);
    SSG_LAUNCH_CHECK("topk_introsort_stream_kernel");
    SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_introsort_kernel<false, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((topk_introsort_kernel<false, NT>), dim3(fallback_blocks(nrows)), dim3(NT), lds, stream, D, rowmax, N, nrows, K,
                       mask_words(N), arena, entry_words(N), rank, tailn, tails, tail_stride_words(tailn), (const int*)redo);
  } else {
  SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_introsort_kernel<LDS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((topk_introsort_kernel<LDS, NT>), dim3(LDS ? nrows : arena_blocks(nrows)), dim3(NT), lds, stream, D, rowmax, N, nrows, K,
                     mask_words(N), arena, entry_words(N), rank, tailn, tails, tail_stride_words(tailn), (const int*)nullptr);
  }
  if (tailn > 0) {
    SSG_LAUNCH_CHECK("topk_introsort_kernel");
    const int head = 64 + tailn;                               // entries a row's tail can cover
    const int wcap = mask_words(head);
    const size_t tlds = ((sizeof(Ctl) + 15) & ~(size_t)15) + (size_t)wcap * (2 * sizeof(uint64_t) + sizeof(uint32_t)) + (size_t)((head + 3) & ~3) * 4;
    if (tlds > 64 * 1024) SSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_introsort_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
    const int per_cu = (int)std::min<size_t>(32, LDS_LIMIT / tlds);
    const int grid = std::min(nrows, 256 * (per_cu > 0 ? per_cu : 1));
    hipLaunchKernelGGL(topk_introsort_tail_kernel, dim3(grid), dim3(64), tlds, stream, tails, tail_stride_words(tailn), rowmax, nrows, K, wcap, rank);
  }
  return SSG_OK;
}

}  // namespace intro
}  // namespace ssg

using namespace ssg;

// workspace = [hand-over records of the tail kernel (split mode) | global arena (only when a row does not fit in LDS)]
// arena = streamed path (default): two left-child buffers per persistent workgroup + the in-place arena of the fallback blocks + one flag
//         per row;  in-place path (SSG_INTRO_STREAM=0, short rows): one row per arena block
static size_t intro_arena_only_bytes(int N, int nrows) {
  if (intro::stream_enabled(N))
    return ((size_t)intro::stream_blocks(N, nrows) * 2 + (size_t)intro::fallback_blocks(nrows)) * intro::entry_words(N) * sizeof(uint32_t) + (size_t)nrows * sizeof(int);
  return (size_t)intro::arena_blocks(nrows) * intro::entry_words(N) * sizeof(uint32_t);
}
extern "C" size_t ssg_topk_rank_introsort_arena_bytes(int N, int nrows) {
  if (N <= 0 || nrows <= 0) return 0;
  return intro::tail_bytes(nrows) + intro_arena_only_bytes(N, nrows);
}
extern "C" size_t ssg_topk_rank_introsort_ws_bytes(int N, int nrows) {
  if (N <= 0 || nrows <= 0) return 0;
  return intro::use_stream(N) ? ssg_topk_rank_introsort_arena_bytes(N, nrows) : intro::tail_bytes(nrows);
}

// byte offset, inside a workspace of ssg_topk_rank_introsort_arena_bytes(N, nrows) bytes, of the per-row int32 flags the streamed kernel
// sets for the rows it hands to the in-place kernel (a pivot inside [0, K)); (size_t)-1 when the streamed kernel is not used for this N.
// Test / diagnostic surface: the flagged rows are the rare path of the streamed replay and get their own check against np.argsort.
extern "C" size_t ssg_topk_rank_introsort_flags_offset(int N, int nrows) {
  if (N <= 0 || nrows <= 0 || !intro::stream_enabled(N)) return (size_t)-1;
  return ssg_topk_rank_introsort_arena_bytes(N, nrows) - (size_t)nrows * sizeof(int);
}

extern "C" int ssg_topk_rank_introsort(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, void* ws,
                                       size_t ws_bytes, hipStream_t stream) {
  if (N < 2 || nrows <= 0 || K <= 0 || K > 64 || K > N || N > (1 << intro::IDX_BITS)) {
    ssg_set_error("ssg_topk_rank_introsort: need 0 < K <= min(64, N), 2 <= N <= 131072 (K=%d N=%d)", K, N);
    return SSG_ERR_INVALID;
  }
  const size_t tb = intro::tail_bytes(nrows), full = ssg_topk_rank_introsort_arena_bytes(N, nrows);
  // rows that do not fit in LDS -- or a caller passing the arena size: that is how rows that fit only ONE per CU get the (faster) streamed
  // kernel: ssg_topk_rank_introsort_ws_bytes() asks for the arena there (use_stream); without it they still run from LDS
  const bool arena = !intro::fits_lds(N) || (ws != nullptr && ws_bytes >= full);
  // LDS-resident rows need no workspace at all: without one (ws == NULL or too small for the hand-over records) the whole replay
  // runs in the single-launch kernel, as before round 3 (slower: one wave finishes each row's tail); only the arena is mandatory
  const bool split = tb > 0 && ws != nullptr && ws_bytes >= (arena ? full : tb);
  const size_t need = arena ? (split ? full : full - tb) : 0;
  if (need > 0 && (ws == nullptr || ws_bytes < need)) {
    ssg_set_error("ssg_topk_rank_introsort: workspace of %zu bytes needed for N=%d, %d rows (got %zu): ssg_topk_rank_introsort_ws_bytes()", need, N, nrows, ws_bytes);
    return SSG_ERR_INVALID;
  }
  uint32_t* tails = split ? (uint32_t*)ws : nullptr;
  uint32_t* ar = arena ? (uint32_t*)((unsigned char*)ws + (split ? tb : 0)) : nullptr;
  // streamed path: [2 buffers per persistent workgroup | fallback arena | flags]
  uint32_t* sbufs = nullptr; int* redo = nullptr;
  if (arena && intro::stream_enabled(N)) {
    sbufs = ar;
    ar = sbufs + (size_t)intro::stream_blocks(N, nrows) * 2 * intro::entry_words(N);
    redo = (int*)(ar + (size_t)intro::fallback_blocks(nrows) * intro::entry_words(N));
  }
  static int nt = -1;
  if (nt < 0) { const char* e_ = getenv("SSG_INTRO_NT"); nt = e_ ? atoi(e_) : 512; }
  int rc;
  if (!arena) {
    rc = nt == 1024 ? intro::launch<true, 1024>(D, rowmax, N, nrows, K, rank, nullptr, tails, stream)
       : nt == 256 ? intro::launch<true, 256>(D, rowmax, N, nrows, K, rank, nullptr, tails, stream)
                   : intro::launch<true, 512>(D, rowmax, N, nrows, K, rank, nullptr, tails, stream);
  } else if (sbufs) {
    rc = intro::stream_nt(N) == 512 ? intro::launch<false, 512>(D, rowmax, N, nrows, K, rank, ar, tails, stream, sbufs, redo)
                    : intro::launch<false, 1024>(D, rowmax, N, nrows, K, rank, ar, tails, stream, sbufs, redo);
  } else {
    rc = nt == 1024 ? intro::launch<false, 1024>(D, rowmax, N, nrows, K, rank, ar, tails, stream)
       : nt == 256 ? intro::launch<false, 256>(D, rowmax, N, nrows, K, rank, ar, tails, stream)
                   : intro::launch<false, 512>(D, rowmax, N, nrows, K, rank, ar, tails, stream);
  }
  if (rc) return rc;
  SSG_LAUNCH_CHECK("topk_introsort_kernel");
  return SSG_OK;
}
