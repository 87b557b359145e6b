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
template <int NL>
__global__ __launch_bounds__(256) void gram_i8_encode_kernel(const float* __restrict__ X, int n, int d, int nkb, int8_t* __restrict__ E,
                                                             long long* __restrict__ norms, int* __restrict__ flag) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* x = X + (int64_t)row * d;
  // panel-major: [row / 64][k block][row % 64][digit][32 k] (see gram_i8_kernel: a tile's stage is one contiguous 64 x 32 NL byte block)
  int8_t* e = E + ((int64_t)(row / GI_T) * nkb * GI_T + (row % GI_T)) * (32 * NL);
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
      p[32 * L] = (int8_t)dg;
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
  const int tiles_n = (N + GI_T - 1) / GI_T, tiles_m = (M + GI_T - 1) / GI_T;
  int tm, tn;
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
    if (tm >= T || tn >= T || tn < tm) return;            // outside the matrix / below the diagonal (diagonal super-blocks only)
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
  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31 -> B row (j), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> A row (i).
  const int gj = tn * GI_T + wn * 32 + l32;
  const bool jok = gj < N;
  const long long nj = jok ? nB[gj] : 0;
  unsigned cmax = 0;
  // mirror: the transposed 32x32 half tile goes through a private LDS patch (the stage buffers are free after the last
  // barrier) so that it is stored as 64-byte row segments like the direct tile, not as scattered 8-byte pieces
  constexpr int MP = 34;                                   // patch pitch in halves (17 dwords: odd)
  hbits* patch = reinterpret_cast<hbits*>(lds) + wave * (32 * MP);
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
      D[(int64_t)li * N + gj] = (hbits)dd;
      cmax = cmax > dd ? cmax : dd;
    }
    if (mirror) patch[l32 * MP + il] = (hbits)dd;
    const unsigned red = halfwave_max(dd);              // row maximum over the 32 columns of this half-wave
    if (l32 == 0 && li < M) atomicMax(&rowmax[li], red);
  }
  if (mirror) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // patch written and read by this wave only
    // lane pair (2j, 2j+1) stores the 32 halves of mirrored row j = this tile's column j: 2 x 32 bytes
    const int mj = lane >> 1, half16 = (lane & 1) * 16;
    const int grow = tn * GI_T + wn * 32 + mj;            // mirrored row (a column of this tile)
    const int gcol = tm * GI_T + wm * 32 + half16;        // first of 16 mirrored columns (rows of this tile)
    if (grow < N) {
      const hbits* src = patch + mj * MP + half16;
      hbits* dst = D + (int64_t)grow * N + gcol;
      if (gcol + 16 <= M && ((((int64_t)grow * N + gcol) & 7) == 0)) {
        uint4 v0, v1;
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = (unsigned)src[2 * e] | ((unsigned)src[2 * e + 1] << 16);
        v0 = make_uint4(w[0], w[1], w[2], w[3]); v1 = make_uint4(w[4], w[5], w[6], w[7]);
        reinterpret_cast<uint4*>(dst)[0] = v0; reinterpret_cast<uint4*>(dst)[1] = v1;
      } else {
        for (int e = 0; e < 16; e++) if (gcol + e < M) dst[e] = src[e];
      }
    }
  }
  if (mirror) {   // mirrored rows are this tile's columns: one lane per column and half-wave
    const unsigned o = (unsigned)__shfl_xor((int)cmax, 32, 64);
    cmax = cmax > o ? cmax : o;
    if (h == 0 && jok) atomicMax(&rowmax[gj], cmax);
  }
}

}  // namespace ssg

using namespace ssg;

extern "C" size_t ssg_gram_i8_encoded_bytes(int n, int d, int ndigits) {       // whole 64-row panels
  return (size_t)((n + GI_T - 1) / GI_T * GI_T) * (size_t)((d + 31) / 32) * 32 * (size_t)ndigits;
}

// Digits + exact squared norms of n rows.  ndigits = 3 (|feat| <= 0.498) or 4 (|feat| <= 1); *flag |= 1 when a half-rounded
// feature does not fit.  E: ssg_gram_i8_encoded_bytes(n, d, ndigits) bytes, norms: n int64.  The caller zeroes *flag first.
extern "C" int ssg_gram_i8_encode(const float* x, int n, int d, int ndigits, void* E, int64_t* norms, int32_t* flag, hipStream_t stream) {
  if (n <= 0 || d <= 0 || d > 16384 || (ndigits != 3 && ndigits != 4)) { ssg_set_error("ssg_gram_i8_encode: need 0 < d <= 16384, ndigits 3 or 4 (n=%d d=%d ndigits=%d)", n, d, ndigits); return SSG_ERR_INVALID; }
  const int nkb = (d + 31) / 32;
  if (ndigits == 3) hipLaunchKernelGGL(gram_i8_encode_kernel<3>, dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, nkb, (int8_t*)E, (long long*)norms, flag);
  else hipLaunchKernelGGL(gram_i8_encode_kernel<4>, dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, nkb, (int8_t*)E, (long long*)norms, flag);
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
  static int kb2 = -1;
  if (kb2 < 0) { const char* e_ = getenv("SSG_I8_KB2"); kb2 = e_ ? atoi(e_) : 1; }   // measured: 1 block per stage (3 waves/SIMD) 2.64 ms, 2 blocks (2 waves/SIMD) 2.9 ms at N=16000
  const bool two = kb2 == 2 && (nkb % 2) == 0;     // two k blocks per LDS stage: half the barriers, but 190 VGPRs
  if (ndigits == 3) { if (two) SSG_GI_LAUNCH(3, 2); else SSG_GI_LAUNCH(3, 1); }
  else SSG_GI_LAUNCH(4, 1);
#undef SSG_GI_LAUNCH
  SSG_LAUNCH_CHECK("gram_i8_kernel");
  return SSG_OK;
}
