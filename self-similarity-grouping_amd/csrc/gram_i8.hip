// gram_i8.hip -- K3 "original distance" (reid/rerank.py:33,61-62) as an EXACT integer Gram on the int8 matrix cores.
//
// The reference rounds the features to half first (feat = input_feature.astype(np.float16), rerank.py:33) and takes
// cdist(feat, feat) in float64.  For |feat| <= 1 (L2-normalised embeddings) every half value is an integer multiple of
// 2^-24, so X = feat * 2^24 is an integer with |X| <= 2^24, every term (x - y)^2 of cdist's sum is a multiple of 2^-48
// below 4, and the float64 running sum never needs more than 51 bits: scipy's squared distance is EXACT.  The same
// exact number comes out of integer arithmetic:
//     d2 * 2^48 = |X_i|^2 + |X_j|^2 - 2 <X_i, X_j>            (int64, < 2^51)
// and the dot product runs on v_mfma_i32_32x32x32_i8 (2x the fp16 rate, 64x the fp64 MFMA rate per multiply) after
// splitting X into NL balanced radix-256 digits  X = sum_a 256^a e_a,  e_a in [-128, 127]:
//     <X, Y> = sum_w 256^w T_w,   T_w = sum_{a+b=w} sum_k ea_k fb_k        (NL^2 digit products, 2 NL - 1 int32 accumulators;
//                                                                         |T_w| <= NL * d * 2^14 < 2^31 for d <= 16384)
// NL = 3 covers |feat| <= 0.498 (every real L2-normalised 2048-d embedding): 9 MFMAs per 32x32x32 block; NL = 4 covers
// |feat| <= 1 (16 MFMAs).  The epilogue is the float64 one of pairwise.hip (sqrt -> half -> square -> half,
// rerank.py:61-62) on the exactly converted integer, so D is bit-identical to the reference by construction (not just
// with high probability) whenever d2 < 32 -- always true for L2-normalised rows (d2 <= 4); for un-normalised features in
// [-1, 1] with larger distances both scipy's running sum and the int64 -> float64 conversion round in the last bit, like
// the fp64 kernel.  Anything larger than 1 (or non-finite) is left to the fp64-MFMA kernel of pairwise.hip: the
// caller picks NL from max|feat|; the encoder additionally raises a device flag if a digit does not fit.
#include "ssg_common.h"
#ifndef SSG_GI_NT_STORE
#define SSG_GI_NT_STORE 0          // A/B knob: the stores of D with the nt cache policy
#endif
#include <cstdlib>

namespace ssg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int GI_T = 64;          // workgroup tile (4 waves, 32x32 outputs each).  Measured alternatives at N=16000 (2.65 ms):
                                  // 128x128 tiles with 16 waves 2.7-2.9 ms, super-block tile order for L2 locality 2.8 ms,
                                  // two k blocks per stage 2.9 ms -- the kernel sits at 46 % MFMA busy either way

// One wave per row: digits of feat*2^24 for every 32-wide k block, and the exact squared norm.  Layout (round 4): PANEL-major,
// [row / 64][k block][row % 64][digit][32 k]: the 64 rows x 32 NL bytes a tile stages per k block are ONE contiguous 6 KB piece (NL = 3)
// instead of 64 pieces of 96 bytes that are 6 KB apart and straddle 128-byte lines -- the texture addresser works through a wave's
// load line by line, and with the row-major layout the global -> register staging alone cost 0.7 of the kernel's 2.7 ms (ablation:
// tools/micro/gram_prof.hip with / without the loads).  Rows beyond n in the last panel are never written: the tile loads clamp to row n - 1.
// PLANAR (round 5, for the LDS-DMA kernel): [row / 64][k block][digit][row % 64][32 k] -- a digit plane of a tile's stage is one contiguous
// 2 KB piece (two 1 KB DMA instructions), and a fragment read is 16 bytes of ONE row of ONE plane (32-byte row pitch).
template <int NL, bool PLANAR>
__global__ __launch_bounds__(256) void gram_i8_encode_kernel(const float* __restrict__ X, int n, int d, int nkb, int8_t* __restrict__ E,
                                                             long long* __restrict__ norms, int* __restrict__ flag) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* x = X + (int64_t)row * d;
  // panel-major: [row / 64][k block][row % 64][digit][32 k] (see gram_i8_kernel: a tile's stage is one contiguous 64 x 32 NL byte block)
  int8_t* e = PLANAR ? E + (int64_t)(row / GI_T) * nkb * (GI_T * 32 * NL) + (row % GI_T) * 32
                     : E + ((int64_t)(row / GI_T) * nkb * GI_T + (row % GI_T)) * (32 * NL);
  long long acc = 0;
  bool bad = false;
  for (int k0 = 0; k0 < nkb * 32; k0 += 64) {
    const int k = k0 + lane;
    if (k >= nkb * 32) break;
    const float v = k < d ? h2f(f2h(x[k])) : 0.f;          // feat = astype(float16)
    if (!(fabsf(v) <= 1.0f)) bad = true;                    // also catches NaN
    const int q = bad ? 0 : (int)(v * 16777216.0f);         // exact: half values <= 1 are multiples of 2^-24
    acc += (long long)q * (long long)q;
    int8_t* p = e + (int64_t)(k >> 5) * (GI_T * 32 * NL) + (k & 31);
    int r = q;
#pragma unroll
    for (int L = 0; L < NL; L++) {
      const int dg = ((r + 128) & 255) - 128;               // balanced digit in [-128, 127]
      p[PLANAR ? GI_T * 32 * L : 32 * L] = (int8_t)dg;
      r = (r - dg) >> 8;
    }
    if (r != 0) bad = true;                                 // |X| beyond NL digits
  }
  for (int sh = 1; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh, 64);
  if (lane == 0) norms[row] = acc;
  if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// One LDS stage of the multiply: KB2 k blocks; As / Bs point at this lane's row of the A / B stage tile (+ its 16-byte k half).
template <int NL, int KB2>
__device__ __forceinline__ void gi_multiply(const unsigned char* As, const unsigned char* Bs, v16i (&acc)[2 * NL - 1]) {
  constexpr int BLK = 32 * NL;
#pragma unroll
  for (int q = 0; q < KB2; q++) {
    v4i a[NL], b[NL];
#pragma unroll
    for (int L = 0; L < NL; L++) { a[L] = *reinterpret_cast<const v4i*>(As + q * BLK + L * 32); b[L] = *reinterpret_cast<const v4i*>(Bs + q * BLK + L * 32); }
    // digit products grouped by weight a+b; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int La = 0; La < NL; La++)
#pragma unroll
      for (int Lb = 0; Lb < NL; Lb++) acc[La + Lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[La], b[Lb], acc[La + Lb], 0, 0, 0);
  }
}

// ---- round 4: the epilogue's sqrt -> half in integer arithmetic.
// The reference's value is d2h(sqrt_f64(d2)) with d2 = U * 2^-48 exact (U = the int64 above).  For d2 < 4 that double rounding equals
// the DIRECT round-to-nearest-even of the exact square root: a half midpoint m is a multiple of 2^-25, m^2 of 2^-50, so unless
// d2 == m^2 exactly |sqrt(d2) - m| >= 2^-50 / (2 m) > 2^-53 * m -- the float64 rounding of the root cannot reach or cross m.  And the
// direct rounding needs no square root beyond a candidate: take h = half(sqrtf((float)U * 2^-48)) (within one half ulp of the answer
// by a wide margin: a relative error of 1e-4 in that root still gives the same result) and compare 4 U with the squares of the two
// midpoints around h, all in int64 (Mi < 2^27, Mi^2 < 2^54): above the upper midpoint -> h + 1, below the lower -> h - 1, exactly on
// one -> the even neighbour.  Checked against numpy on 3.6 M values incl. every midpoint square +-3 (tools: DESIGN.md section 9).
// ~45 integer / float32 instructions instead of ~150 float64 ones (int64 -> float64, sqrt, the branchy direct double -> half).
__device__ __forceinline__ long long half_units24(unsigned h) {      // value of the non-negative half h in units of 2^-24
  const unsigned e = h >> 10, m = h & 1023u;
  return e == 0 ? (long long)m : (long long)((unsigned long long)(1024u | m) << (e - 1));
}
__device__ __forceinline__ hbits sqrt_units48_to_half(long long u) {  // 0 <= u < 2^50
  const float x = (float)u * 3.5527136788005009e-15f;                 // 2^-48: exact scaling, the conversion rounds (candidate only)
  unsigned h = (unsigned)f2h(__builtin_amdgcn_sqrtf(x));
  const long long vh = half_units24(h);
  const long long mu = vh + half_units24(h + 1u), ml = (h ? half_units24(h - 1u) : 0) + vh;
  const long long x4 = u << 2, mu2 = mu * mu, ml2 = ml * ml;
  const bool odd = (h & 1u) != 0;
  const bool inc = x4 > mu2 || (x4 == mu2 && odd);
  const bool dec = h != 0 && (x4 < ml2 || (x4 == ml2 && odd));
  return (hbits)(h + (inc ? 1u : 0u) - (dec ? 1u : 0u));
}
// max over the 32 lanes of each half-wave (valid in lanes 0 and 32): four DPP steps inside the 16-lane rows + one cross-row exchange
__device__ __forceinline__ unsigned halfwave_max(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); v = v > o ? v : o;     // quad_perm [1,0,3,2]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); v = v > o ? v : o;     // quad_perm [2,3,0,1]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); v = v > o ? v : o;    // row_ror:4
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); v = v > o ? v : o;    // row_ror:8
  o = (unsigned)__shfl_xor((int)v, 16, 64); v = v > o ? v : o;
  return v;
}

// tile (tm, tn) of this workgroup: row-major over the row block, or -- one GPU holding the whole matrix -- the upper triangle, in
// super-blocks of sb x sb tiles per XCD.  false: the workgroup has no tile (padding of the super-block grid).
__device__ __forceinline__ bool gi_pick_tile(int M, int N, int symmetric, int sb, int& tm, int& tn) {
  const int tiles_n = (N + GI_T - 1) / GI_T, tiles_m = (M + GI_T - 1) / GI_T;
  if ((symmetric & 1) && sb > 0) {
    // round 4: the upper triangle in SUPER-BLOCKS of sb x sb tiles, each XCD working through whole super-blocks.  In the row-major
    // order below, the ~96 tiles an XCD runs at a time are consecutive columns of ONE tile row: they share the A panel, but every tile
    // pulls its own B panel (393 KB of digit planes) through the fabric -- 12.6 GB per launch for 98 MB of planes, at the ~4.8 TB/s the
    // L2 <-> fabric path gives, which is the kernel's time.  The tiles of a super-block need sb A panels + sb B panels, and they walk
    // k together, so the L2 holds the current window of each: 2 sb panel streams per sb^2 tiles instead of sb^2 + sb.  Measured (PMC
    // FETCH_SIZE, sb = 10): 12.6 -> 6.5 GB out of the L2s per launch -- and the SAME time within the box-to-box spread (2.6-2.7 ms): the
    // fabric is not what holds this kernel at 47 % matrix-pipe busy either.
    const int T = tiles_n, nb = (T + sb - 1) / sb;
    const int t = xcd_remap((int)blockIdx.x, nb * (nb + 1) / 2 * sb * sb);
    const int blk = t / (sb * sb), within = t - blk * (sb * sb);
    int r = (int)(((2.0 * nb + 1.0) - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)blk)) * 0.5);
    while (r > 0 && r * nb - r * (r - 1) / 2 > blk) r--;
    while ((r + 1) * nb - (r + 1) * r / 2 <= blk) r++;
    const int c = r + (blk - (r * nb - r * (r - 1) / 2));
    tm = r * sb + within / sb; tn = c * sb + within % sb;
    if (tm >= T || tn >= T || tn < tm) return false;           // outside the matrix / below the diagonal (diagonal super-blocks only)
  } else if (symmetric & 1) {
    const int T = tiles_n;
    const int t = xcd_remap((int)blockIdx.x, T * (T + 1) / 2);
    int r = (int)(((2.0 * T + 1.0) - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)t)) * 0.5);
    while (r > 0 && r * T - r * (r - 1) / 2 > t) r--;
    while ((r + 1) * T - (r + 1) * r / 2 <= t) r++;
    tm = r; tn = r + (t - (r * T - r * (r - 1) / 2));
  } else {
    const int tile = xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
    tm = tile / tiles_n; tn = tile % tiles_n;
  }
  return true;
}

// sqrt -> half -> square -> half of the exact integer distance, D store (+ the mirrored half tile through an LDS patch), row maxima
template <int NL>
__device__ __forceinline__ void gi_epilogue(v16i (&acc)[2 * NL - 1], unsigned char* lds, const long long* __restrict__ nA, const long long* __restrict__ nB, int M, int N,
                                            int rowA0, hbits* __restrict__ D, unsigned* __restrict__ rowmax, int symmetric, int tm, int tn, bool mirror) {
  constexpr int NACC = 2 * NL - 1;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31 -> B row (j), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> A row (i).
  const int gj = tn * GI_T + wn * 32 + l32;
  const bool jok = gj < N;
  const long long nj = jok ? nB[gj] : 0;
  unsigned cmax = 0;
  // Both the tile and (mirror) its transpose go through private LDS patches of the wave (the stage buffers are free after the last
  // barrier) and leave as 32-byte row pieces, two 16-byte stores per lane.  (Rounds 1-4 stored the direct tile straight from the
  // accumulators: 16 store instructions of 2 bytes per lane, and took every row maximum with a 5-step DPP reduction per accumulator
  // register; the row pieces a lane pair stores now hold a whole row of the wave tile, so its maximum is 15 compares + one exchange.)
  constexpr int MP = 34;                                   // patch pitch in halves (17 dwords: odd)
  hbits* patch = reinterpret_cast<hbits*>(lds) + wave * (32 * MP);            // 2176 B per wave: first [row][col], then (mirror) [col][row]
  unsigned dds[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int il = (r & 3) + 8 * (r >> 2) + 4 * h;          // row of this accumulator element inside the wave tile
    const int li = tm * GI_T + wm * 32 + il;
    const bool ok = jok && li < M;
    unsigned dd = 0;
    if (ok) {
      long long dot = 0;
#pragma unroll
      for (int w = NACC - 1; w >= 0; w--) dot = dot * 256 + (long long)acc[w][r];
      long long d2i = nA[li] + nj - 2 * dot;           // exact squared distance in units of 2^-48
      if (d2i < 0 || rowA0 + li == gj) d2i = 0;         // cannot be negative; cdist(x, x) diagonal is exactly 0
      if (d2i < (1LL << 50) && !(symmetric & 2)) {
        const hbits hh = sqrt_units48_to_half(d2i);      // == d2h(sqrt((double)d2i * 2^-48)) for d2 < 4: every L2-normalised pair
        dd = h_mul(hh, hh);                              // np.power(half, 2) rerank.py:62
      } else {
        const double s = (double)d2i * 3.5527136788005009e-15;   // 2^-48, exact (d2i < 2^53)
        const double sq = sqrt(s);
        const hbits hh = d2h(sq);                         // cdist(...).astype(float16)   rerank.py:61
        // np.power(half, 2) rerank.py:62; MemorySave branch (:49-59): np.power(cdist, 2).astype(float16), one rounding
        dd = (symmetric & 2) ? d2h(sq * sq) : h_mul(hh, hh);
      }
      cmax = cmax > dd ? cmax : dd;
    }
    dds[r] = dd;
    patch[il * MP + l32] = (hbits)dd;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the patch is written and read by this wave only
  // lane pair (2j, 2j+1) stores the 32 halves of row j of the wave tile (then of mirrored row j = this tile's column j): 2 x 32 bytes
  const int mj = lane >> 1, half16 = (lane & 1) * 16;
  auto store_row = [&](const hbits* src, int grow, int gcol, int nrow_lim, int ncol_lim) {
    if (grow < nrow_lim) {
      hbits* dst = D + (int64_t)grow * N + gcol;
      if (gcol + 16 <= ncol_lim && ((((int64_t)grow * N + gcol) & 7) == 0)) {
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = (unsigned)src[2 * e] | ((unsigned)src[2 * e + 1] << 16);
#if SSG_GI_NT_STORE
        { typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
          const v4u_ s0_ = {w[0], w[1], w[2], w[3]}, s1_ = {w[4], w[5], w[6], w[7]};
          __builtin_nontemporal_store(s0_, reinterpret_cast<v4u_*>(dst)); __builtin_nontemporal_store(s1_, reinterpret_cast<v4u_*>(dst) + 1); }
#else
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]); reinterpret_cast<uint4*>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
#endif
      } else {
        for (int e = 0; e < 16; e++) if (gcol + e < ncol_lim) dst[e] = src[e];
      }
    }
  };
  {
    const int li = tm * GI_T + wm * 32 + mj;               // row of the tile, columns gcol .. gcol + 15
    const hbits* src = patch + mj * MP + half16;
    store_row(src, li, tn * GI_T + wn * 32 + half16, M, N);
    unsigned rmax = 0;                                      // (entries outside the matrix are 0 in the patch)
#pragma unroll
    for (int e = 0; e < 16; e++) { const unsigned v = src[e]; rmax = rmax > v ? rmax : v; }
    const unsigned o = (unsigned)__shfl_xor((int)rmax, 1, 64);
    rmax = rmax > o ? rmax : o;
    if ((lane & 1) == 0 && li < M) atomicMax(&rowmax[li], rmax);
  }
  if (mirror) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // every lane is done reading the [row][col] patch
#pragma unroll
    for (int r = 0; r < 16; r++) patch[l32 * MP + (r & 3) + 8 * (r >> 2) + 4 * h] = (hbits)dds[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    store_row(patch + mj * MP + half16, tn * GI_T + wn * 32 + mj, tm * GI_T + wm * 32 + half16, N, M);
    // mirrored rows are this tile's columns: one lane per column and half-wave
    const unsigned o = (unsigned)__shfl_xor((int)cmax, 32, 64);
    cmax = cmax > o ? cmax : o;
    if (h == 0 && jok) atomicMax(&rowmax[gj], cmax);
  }
}

// D[i,j] = half(half(sqrt(d2))^2) for rows [rowA0, rowA0+M) x all N columns, atomicMax rowmax.  EA = encoded rows of the
// row block, EB = encoded rows of the whole set.  symmetric: only tiles on/above the diagonal are launched, mirrored on store.
template <int NL, int KB2>
__global__ __launch_bounds__(256, 2) void gram_i8_kernel(const int8_t* __restrict__ EA, const int8_t* __restrict__ EB,
                                                         const long long* __restrict__ nA, const long long* __restrict__ nB, int M, int N, int nkb,
                                                         int rowA0, hbits* __restrict__ D, unsigned* __restrict__ rowmax, int symmetric,
                                                         const int* __restrict__ flag, int sb) {
  if (*flag) return;     // a feature did not fit NL digits: the caller falls back to the fp64 kernel
  constexpr int BLK = 32 * NL;          // bytes of one k block of one row
  constexpr int SB = KB2 * BLK;         // bytes of one stage row: KB2 consecutive k blocks (one barrier per KB2 * NL^2 MFMAs)
  constexpr int PITCH = SB + 16;        // LDS row pitch: pitch/16 odd (7, 9, 13, 17) -> conflict-free b128
  constexpr int CPR = SB / 16;          // 16-byte chunks per row and stage
  constexpr int NCH = (2 * GI_T * CPR) / 256;   // chunks per thread: A tile + B tile
  constexpr int NACC = 2 * NL - 1;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * GI_T * PITCH];
  int tm, tn;
  if (!gi_pick_tile(M, N, symmetric, sb, tm, tn)) return;
  const bool mirror = (symmetric & 1) && tn > tm;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
  // staging: the A tile and the B tile together are 128 rows x CPR chunks of 16 bytes = NL chunks per thread
  const uint4* gp[NCH]; int lo[NCH];
#pragma unroll
  for (int p = 0; p < NCH; p++) {
    const int c = tid + 256 * p, mat = c / (GI_T * CPR), rem = c - mat * (GI_T * CPR), row = rem / CPR, ch = rem - row * CPR;
    // global row of the whole set (A rows: the row block starts at rowA0; E is the ONE encoded table of all N rows, panel-major)
    const int grow = mat ? min(tn * GI_T + row, N - 1) : rowA0 + min(tm * GI_T + row, M - 1);
    constexpr int CPB = BLK / 16;                      // 16-byte chunks of one k block of one row; chunk ch of the stage row sits in k block ch / CPB
    gp[p] = reinterpret_cast<const uint4*>(EB + ((int64_t)(grow / GI_T) * nkb * GI_T + (grow % GI_T)) * BLK) + (ch / CPB) * (GI_T * CPB) + ch % CPB;
    lo[p] = mat * (GI_T * PITCH) + row * PITCH + ch * 16;
  }
  v16i acc[NACC];
#pragma unroll
  for (int w = 0; w < NACC; w++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[w][r] = 0;

  const int nst = nkb / KB2;
  const unsigned char* As0 = lds + (wm * 32 + l32) * PITCH + h * 16;
  const unsigned char* Bs0 = lds + GI_T * PITCH + (wn * 32 + l32) * PITCH + h * 16;
  // A stage (18 MFMAs with 3 digits) is shorter than the L2 latency, so the global loads run TWO stages ahead in two
  // register sets (set 0 holds odd stages, set 1 even ones); stage indices are clamped instead of branching around
  // loads.  Named scalars, not arrays: hipcc keeps loop-carried uint4 arrays in scratch memory.
  static_assert(NCH <= 6, "staging code is written for at most 6 chunks per thread");
  uint4 pa0, pa1, pa2, pa3, pa4, pa5, pb0, pb1, pb2, pb3, pb4, pb5;
  pa0 = pa1 = pa2 = pa3 = pa4 = pa5 = pb0 = pb1 = pb2 = pb3 = pb4 = pb5 = make_uint4(0, 0, 0, 0);
#define SSG_GL(S, ST)                                                                                 \
  {                                                                                                   \
    const int so_ = (ST) * (GI_T * CPR);            /* a stage = KB2 k blocks of the panel, 64 rows each */ \
    p##S##0 = gp[0][so_]; if (NCH > 1) p##S##1 = gp[1][so_]; if (NCH > 2) p##S##2 = gp[2][so_];         \
    if (NCH > 3) p##S##3 = gp[3][so_]; if (NCH > 4) p##S##4 = gp[4][so_]; if (NCH > 5) p##S##5 = gp[5][so_]; \
  }
#define SSG_LS(S, BUF)                                                                                \
  {                                                                                                   \
    unsigned char* b_ = lds + (BUF) * (2 * GI_T * PITCH);                                             \
    *reinterpret_cast<uint4*>(b_ + lo[0]) = p##S##0;                                                  \
    if (NCH > 1) *reinterpret_cast<uint4*>(b_ + lo[1]) = p##S##1;                                     \
    if (NCH > 2) *reinterpret_cast<uint4*>(b_ + lo[2]) = p##S##2;                                     \
    if (NCH > 3) *reinterpret_cast<uint4*>(b_ + lo[3]) = p##S##3;                                     \
    if (NCH > 4) *reinterpret_cast<uint4*>(b_ + lo[4]) = p##S##4;                                     \
    if (NCH > 5) *reinterpret_cast<uint4*>(b_ + lo[5]) = p##S##5;                                     \
  }
#ifdef SSG_GI_NO_LOADS          // ablation (tools/micro/gram_prof.hip): stage 0 only, the loop re-stores the same registers
#undef SSG_GL
#define SSG_GL(S, ST) { if ((ST) == 0 && st_dummy_ == 0) { p##S##0 = gp[0][0]; if (NCH > 1) p##S##1 = gp[1][0]; if (NCH > 2) p##S##2 = gp[2][0]; } }
  const int st_dummy_ = 0;
#endif
  SSG_GL(a, 0)
  SSG_LS(a, 0)
  { const int s1 = min(1, nst - 1); SSG_GL(a, s1) }
  __syncthreads();
  int st = 0;
  for (; st + 1 < nst; st += 2) {
    { const int sn = min(st + 2, nst - 1); SSG_GL(b, sn) }
    __builtin_amdgcn_sched_barrier(0);
    gi_multiply<NL, KB2>(As0, Bs0, acc);
    SSG_LS(a, 1)                                   // stage st+1 -> buffer 1
    __syncthreads();
    { const int sn = min(st + 3, nst - 1); SSG_GL(a, sn) }
    __builtin_amdgcn_sched_barrier(0);
    gi_multiply<NL, KB2>(As0 + 2 * GI_T * PITCH, Bs0 + 2 * GI_T * PITCH, acc);
    SSG_LS(b, 0)                                   // stage st+2 -> buffer 0
    __syncthreads();
  }
  if (st < nst) { gi_multiply<NL, KB2>(As0, Bs0, acc); __syncthreads(); }      // odd stage count: the last stage sits in buffer 0
#undef SSG_GL
#undef SSG_LS

#ifdef SSG_GI_NO_EPI            // ablation: one store per lane keeps the accumulators alive
  { int sum_ = 0;
#pragma unroll
    for (int w = 0; w < NACC; w++)
#pragma unroll
      for (int r = 0; r < 16; r++) sum_ += acc[w][r];
    if (sum_ == 0x7fffffff) D[0] = (hbits)sum_;
    return; }
#endif
  gi_epilogue<NL>(acc, lds, nA, nB, M, N, rowA0, D, rowmax, symmetric, tm, tn, mirror);
}


// ---- round 5: the same tile with LDS-DMA stages on digit PLANES -----------------------------------------------------------------------
// The register-staged kernel above holds two stages' worth of global loads in 48 VGPRs (150 in all: three waves per SIMD) and pays the
// VGPR -> LDS transfer of every byte (ds_write_b128: ~79 B/clk per CU, next to the fragment reads).  Here a stage goes global -> LDS
// directly (buffer_load ... lds, 16 bytes per lane, 1 KB per wave instruction): no staging registers, no ds_write; NS stages of
// [A: NL planes x 2 KB][B: NL planes x 2 KB] in separate __shared__ arrays, loads NS - 1 k blocks ahead.  The planar layout of the
// encoder makes a plane of a stage ONE contiguous 2 KB piece (64 rows x 32 bytes) = two DMA instructions; rows land unpadded (the DMA is
// lane-linear), and the two 16-byte halves of a row are swapped for rows with bit 3 set -- on the global side (a lane fetches the other
// half) and again in the fragment reads -- so that the 16 lanes of a ds_read_b128 group (rows r .. r+3, r+12 .. r+15, r+20 .. r+27) hit
// 64 distinct banks.  Same MFMA sequence, accumulators and epilogue as the kernel above: D is bit-identical.
#define SSG_GI_LDSP(ptr_) ((__attribute__((address_space(3))) void*)(ptr_))
__device__ __forceinline__ void gi_dma3(__amdgpu_buffer_rsrc_t rs, unsigned char* l0, unsigned char* l1, unsigned char* l2, unsigned g0, unsigned g1, unsigned g2) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, SSG_GI_LDSP(l0), 16, g0, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, SSG_GI_LDSP(l1), 16, g1, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, SSG_GI_LDSP(l2), 16, g2, 0, 0, 0);
}
template <int NL, int NS>
__global__ __launch_bounds__(256, NS == 3 ? 4 : 3) void gram_i8_dma_kernel(const int8_t* __restrict__ E, unsigned e_bytes, const long long* __restrict__ nA, const long long* __restrict__ nB,
                                                             int M, int N, int nkb, int rowA0, hbits* __restrict__ D, unsigned* __restrict__ rowmax, int symmetric,
                                                             const int* __restrict__ flag, int sb) {
  if (*flag) return;
  constexpr int PLANE = GI_T * 32;                 // bytes of one digit plane of one operand and k block
  constexpr int STAGE = 2 * NL * PLANE;            // [A planes][B planes]
  constexpr int NACC = 2 * NL - 1;
  constexpr int NDMA = 2 * NL * 2;                 // 1 KB DMA instructions per stage (two per plane)
  static_assert(NDMA % 4 == 0, "the four waves share a stage's DMA instructions evenly");
  constexpr int TDMA = NDMA / 4;                   // per wave and stage
  static_assert(NS == 3 || NS == 4, "three or four stages");
  __shared__ __attribute__((aligned(1024))) unsigned char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st2[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st3[NS == 4 ? STAGE : 1024];
  int tm, tn;
  if (!gi_pick_tile(M, N, symmetric, sb, tm, tn)) return;
  const bool mirror = (symmetric & 1) && tn > tm;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
  // ---- DMA addressing: instruction q = wave * TDMA + j of a stage fills operand q / (2 NL), plane (q % (2 NL)) / 2, rows 32 (q & 1) + lane / 2
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(E), 0, e_bytes, 0x00020000);
  // planar layout: [panel][k block][digit][row % 64][32].  A lane's LDS slot is row = 32 (q & 1) + lane / 2 of the plane, 16-byte half
  // `slot`; rows are clamped PER LANE (grow), so a slot may hold another global row's bytes for rows beyond the matrix (their outputs
  // are never stored) -- the swap bit of the half follows the LDS row, not the global row
  unsigned goff[TDMA]; int loff[TDMA];
#pragma unroll
  for (int j = 0; j < TDMA; j++) {
    const int q = wave * TDMA + j, op = q / (2 * NL), pl = (q % (2 * NL)) >> 1, row = 32 * (q & 1) + (lane >> 1), slot = lane & 1;
    const int grow = op ? min(tn * GI_T + row, N - 1) : rowA0 + min(tm * GI_T + row, M - 1);
    goff[j] = (unsigned)(((int64_t)(grow / GI_T) * nkb * NL + pl) * PLANE + (grow % GI_T) * 32 + ((slot ^ ((row >> 3) & 1)) * 16));
    loff[j] = op * (NL * PLANE) + pl * PLANE + (q & 1) * 1024;           // (+ 16 * lane: implicit in the DMA)
  }
  const unsigned kstep = (unsigned)(NL * PLANE);                          // bytes from k block kb to kb + 1 inside a panel
  int dn = 0;
  static_assert(TDMA == 3, "gi_dma3 issues three instructions per wave and stage (three digits)");
  // (the instructions live in a NON-template device function: with the builtin inside this kernel template hipcc's HOST pass silently
  //  dropped the kernel's launch stub -- the library then failed to load with an undefined __device_stub__ symbol)
#define SSG_GI_DMA(ST)                                                                                                \
  { const unsigned kb_ = (unsigned)dn * kstep;                                                                        \
    gi_dma3(rs, ST + loff[0], ST + loff[1], ST + loff[2], goff[0] + kb_, goff[1] + kb_, goff[2] + kb_);                \
    dn += dn < nkb - 1 ? 1 : 0; }                 /* the tail re-fetches the last block (keeps the vmcnt accounting uniform) */
  // ---- fragment addressing: lane (row l32 of its wave's 32-row block, k half h)
  const int arow = wm * 32 + l32, brow = wn * 32 + l32;
  const int aoff = arow * 32 + ((h ^ ((arow >> 3) & 1)) * 16), boff = NL * PLANE + brow * 32 + ((h ^ ((brow >> 3) & 1)) * 16);
#define SSG_GI_MMA(ST)                                                                                                \
  { v4i a_[NL], b_[NL];                                                                                               \
    _Pragma("unroll") for (int L = 0; L < NL; L++) {                                                                 \
      a_[L] = *reinterpret_cast<const v4i*>(ST + aoff + L * PLANE); b_[L] = *reinterpret_cast<const v4i*>(ST + boff + L * PLANE); } \
    _Pragma("unroll") for (int La = 0; La < NL; La++) _Pragma("unroll") for (int Lb = 0; Lb < NL; Lb++)              \
      acc[La + Lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_[La], b_[Lb], acc[La + Lb], 0, 0, 0); }
  v16i acc[NACC];
#pragma unroll
  for (int w = 0; w < NACC; w++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[w][r] = 0;
  // publish = k block t has landed for every wave AND every wave is done reading the stage that is refilled next (lgkmcnt(0): hipcc sinks
  // the fragment reads' waits below a bare barrier -- the race conv_dma_kernel had in rounds 1-2)
#define SSG_GI_PUBLISH() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TDMA * (NS - 2)) : "memory")
  SSG_GI_DMA(st0)
  SSG_GI_DMA(st1)
  if constexpr (NS == 4) SSG_GI_DMA(st2)
  const int nfull = nkb / NS * NS;
  for (int kt = 0; kt < nfull; kt += NS) {
    if constexpr (NS == 3) {
      SSG_GI_PUBLISH(); SSG_GI_DMA(st2) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st0)
      SSG_GI_PUBLISH(); SSG_GI_DMA(st0) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st1)
      SSG_GI_PUBLISH(); SSG_GI_DMA(st1) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st2)
    } else {
      SSG_GI_PUBLISH(); SSG_GI_DMA(st3) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st0)
      SSG_GI_PUBLISH(); SSG_GI_DMA(st0) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st1)
      SSG_GI_PUBLISH(); SSG_GI_DMA(st1) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st2)
      SSG_GI_PUBLISH(); SSG_GI_DMA(st2) __builtin_amdgcn_sched_barrier(0); SSG_GI_MMA(st3)
    }
  }
  if (nkb - nfull >= 1) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_GI_MMA(st0) }
  if (nkb - nfull >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_GI_MMA(st1) }
  if constexpr (NS == 4) { if (nkb - nfull == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_GI_MMA(st2) } }
  __syncthreads();                      // drains the (clamped, redundant) tail DMAs before stage 0 becomes the mirror patches
#undef SSG_GI_DMA
#undef SSG_GI_MMA
#undef SSG_GI_PUBLISH
  gi_epilogue<NL>(acc, st0, nA, nB, M, N, rowA0, D, rowmax, symmetric, tm, tn, mirror);
}

}  // namespace ssg

using namespace ssg;

// which kernel (and therefore which digit layout) a problem gets: the LDS-DMA kernel for 3 digits when the encoded table fits one buffer
// resource; SSG_I8_DMA=0: the register-staged kernel of rounds 1-4.  The switch is read ONCE per process: the encoder and the multiply
// are two calls and must agree on the layout (a change of the variable between them silently mismatched the two).
static bool gi_use_dma(int n, int d, int ndigits) {
  static int env_on = -1;
  if (env_on < 0) { const char* e_ = getenv("SSG_I8_DMA"); env_on = (e_ ? atoi(e_) : 1) != 0 ? 1 : 0; }
  const size_t bytes = (size_t)((n + GI_T - 1) / GI_T * GI_T) * (size_t)((d + 31) / 32) * 32 * (size_t)ndigits;
  return env_on != 0 && ndigits == 3 && bytes < 0xfffffff0ull && (d + 31) / 32 >= 4;
}

extern "C" size_t ssg_gram_i8_encoded_bytes(int n, int d, int ndigits) {       // whole 64-row panels
  return (size_t)((n + GI_T - 1) / GI_T * GI_T) * (size_t)((d + 31) / 32) * 32 * (size_t)ndigits;
}

// Digits + exact squared norms of n rows.  ndigits = 3 (|feat| <= 0.498) or 4 (|feat| <= 1); *flag |= 1 when a half-rounded
// feature does not fit.  E: ssg_gram_i8_encoded_bytes(n, d, ndigits) bytes, norms: n int64.  The caller zeroes *flag first.
extern "C" int ssg_gram_i8_encode(const float* x, int n, int d, int ndigits, void* E, int64_t* norms, int32_t* flag, hipStream_t stream) {
  if (n <= 0 || d <= 0 || d > 16384 || (ndigits != 3 && ndigits != 4)) { ssg_set_error("ssg_gram_i8_encode: need 0 < d <= 16384, ndigits 3 or 4 (n=%d d=%d ndigits=%d)", n, d, ndigits); return SSG_ERR_INVALID; }
  const int nkb = (d + 31) / 32;
  if (ndigits == 3 && gi_use_dma(n, d, ndigits))
    hipLaunchKernelGGL((gram_i8_encode_kernel<3, true>), dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, nkb, (int8_t*)E, (long long*)norms, flag);
  else if (ndigits == 3) hipLaunchKernelGGL((gram_i8_encode_kernel<3, false>), dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, nkb, (int8_t*)E, (long long*)norms, flag);
  else hipLaunchKernelGGL((gram_i8_encode_kernel<4, false>), dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, nkb, (int8_t*)E, (long long*)norms, flag);
  SSG_LAUNCH_CHECK("gram_i8_encode_kernel");
  return SSG_OK;
}

// Same contract as ssg_sqdist_self_f16 (rows [row0,row0+nrows) x N of the half original distance + row maxima) from the
// encoded features; does nothing when *flag != 0 (the caller then runs ssg_sqdist_self_f16).
extern "C" int ssg_sqdist_self_i8(const void* E, const int64_t* norms, int N, int d, int ndigits, int row0, int nrows, int memory_save, uint16_t* D, uint32_t* rowmax,
                                  const int32_t* flag, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > N || d <= 0 || d > 16384 || (ndigits != 3 && ndigits != 4)) {
    ssg_set_error("ssg_sqdist_self_i8: bad shape N=%d d=%d ndigits=%d row0=%d nrows=%d", N, d, ndigits, row0, nrows);
    return SSG_ERR_INVALID;
  }
  const int nkb = (d + 31) / 32;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream, rowmax, nrows, 0u);
  const int symmetric = (row0 == 0 && nrows == N) ? 1 : 0;
  const int T = (N + GI_T - 1) / GI_T;
  static int sbv = -1;
  if (sbv < 0) { const char* e_ = getenv("SSG_I8_SB"); sbv = e_ ? atoi(e_) : 6; if (sbv < 0 || sbv > 64) sbv = 0; }      // super-block edge in tiles (0: row-major order); measured at N = 16 000: 0 / 6 / 8 / 10 / 12 / 16 / 24 -> 2.68 / 2.60 / 2.76 / 2.72 / 2.70 / 2.82 / 3.14 ms
  const int sb = symmetric ? sbv : 0;
  const int64_t nbk = sb ? (T + sb - 1) / sb : 0;
  const int64_t tiles = symmetric ? (sb ? nbk * (nbk + 1) / 2 * sb * sb : (int64_t)T * (T + 1) / 2) : (int64_t)((nrows + GI_T - 1) / GI_T) * T;
  if (tiles > 0x7fffffff) { ssg_set_error("ssg_sqdist_self_i8: too many tiles"); return SSG_ERR_INVALID; }
  const int8_t* e = (const int8_t*)E;
#define SSG_GI_LAUNCH(NL_, KB_) hipLaunchKernelGGL((gram_i8_kernel<NL_, KB_>), dim3((unsigned)tiles), dim3(256), 0, stream, e, e, \
    (const long long*)norms + row0, (const long long*)norms, nrows, N, nkb, row0, D, rowmax, symmetric | (memory_save ? 2 : 0), flag, sb)
  if (gi_use_dma(N, d, ndigits)) {
    const unsigned e_bytes = (unsigned)ssg_gram_i8_encoded_bytes(N, d, ndigits);
#define SSG_GI_DMA_LAUNCH(NS_) hipLaunchKernelGGL((gram_i8_dma_kernel<3, NS_>), dim3((unsigned)tiles), dim3(256), 0, stream, e, e_bytes, (const long long*)norms + row0, \
      (const long long*)norms, nrows, N, nkb, row0, D, rowmax, symmetric | (memory_save ? 2 : 0), flag, sb)
    const char* s_ = getenv("SSG_I8_DMA_STAGES");      // 3 (default): 36 KB, four workgroups per CU; 4: 48 KB, three
    if (s_ && atoi(s_) == 4) SSG_GI_DMA_LAUNCH(4); else SSG_GI_DMA_LAUNCH(3);
#undef SSG_GI_DMA_LAUNCH
    SSG_LAUNCH_CHECK("gram_i8_dma_kernel");
    return SSG_OK;
  }
  static int kb2 = -1;
  if (kb2 < 0) { const char* e_ = getenv("SSG_I8_KB2"); kb2 = e_ ? atoi(e_) : 1; }   // measured: 1 block per stage (3 waves/SIMD) 2.64 ms, 2 blocks (2 waves/SIMD) 2.9 ms at N=16000
  const bool two = kb2 == 2 && (nkb % 2) == 0;     // two k blocks per LDS stage: half the barriers, but 190 VGPRs
  if (ndigits == 3) { if (two) SSG_GI_LAUNCH(3, 2); else SSG_GI_LAUNCH(3, 1); }
  else SSG_GI_LAUNCH(4, 1);
#undef SSG_GI_LAUNCH
  SSG_LAUNCH_CHECK("gram_i8_kernel");
  return SSG_OK;
}
