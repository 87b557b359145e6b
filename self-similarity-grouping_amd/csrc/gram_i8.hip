// gram_i8.hip -- K3 "original distance" (reid/rerank.py:33,61-62) as an EXACT integer Gram on the int8 matrix cores.
//
// The reference rounds the features to half first (feat = input_feature.astype(np.float16), rerank.py:33) and takes
// cdist(feat, feat) in float64.  For |feat| <= 1 (L2-normalised embeddings) every half value is an integer multiple of
// 2^-24, so X = feat * 2^24 is an integer with |X| <= 2^24, every term (x - y)^2 of cdist's sum is a multiple of 2^-48
// below 4, and the float64 running sum never needs more than 51 bits: scipy's squared distance is EXACT.  The same
// exact number comes out of integer arithmetic:
//     d2 * 2^48 = |X_i|^2 + |X_j|^2 - 2 <X_i, X_j>            (int64, < 2^51)
// and the dot product runs on v_mfma_i32_32x32x32_i8 (2x the fp16 rate, 64x the fp64 MFMA rate per multiply) after
// splitting X into four balanced radix-128 digits  X = d0 + 128 d1 + 128^2 d2 + 128^3 d3,  d0..2 in [-64, 63]:
//     <X, Y> = sum_w 128^w T_w,   T_w = sum_{a+b=w} sum_k da_k eb_k        (16 digit products, 7 int32 accumulators;
//                                                                         |T_w| <= 4 * 2048 * 64 * 64 < 2^26 per 2048 terms)
// The epilogue is the float64 one of pairwise.hip (sqrt -> half -> square -> half, rerank.py:61-62) on the exactly
// converted integer, so D is bit-identical to the reference by construction (not just with high probability).
// A feature outside [-1, 1] (or non-finite) raises a device flag: the int8 kernel then does nothing and the caller runs
// the fp64-MFMA kernel of pairwise.hip instead.
#include "ssg_common.h"

namespace ssg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int GI_T = 64;          // workgroup tile (4 waves, 32x32 outputs each)
constexpr int GI_PITCH = 144;     // LDS bytes per row of one 32-wide k block: 4 digits x 32 B + 16 B pad (pitch/16 odd)

// One wave per row: digits of feat*2^24 for every 32-wide k block, laid out [row][k block][digit][32 k] (128 B per
// block: the layout both the global tile loads and the LDS fragment reads use), and the exact squared norm.
__global__ __launch_bounds__(256) void gram_i8_encode_kernel(const float* __restrict__ X, int n, int d, int nkb, int8_t* __restrict__ E,
                                                             long long* __restrict__ norms, int* __restrict__ flag) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* x = X + (int64_t)row * d;
  int8_t* e = E + (int64_t)row * nkb * 128;
  long long acc = 0;
  bool bad = false;
  for (int k0 = 0; k0 < nkb * 32; k0 += 64) {
    const int k = k0 + lane;
    if (k >= nkb * 32) break;
    const float v = k < d ? h2f(f2h(x[k])) : 0.f;          // feat = astype(float16)
    if (!(fabsf(v) <= 1.0f)) bad = true;                    // also catches NaN
    const int q = bad ? 0 : (int)(v * 16777216.0f);         // exact: half values <= 1 are multiples of 2^-24
    acc += (long long)q * (long long)q;
    const int d0 = ((q + 64) & 127) - 64; const int q1 = (q - d0) >> 7;
    const int d1 = ((q1 + 64) & 127) - 64; const int q2 = (q1 - d1) >> 7;
    const int d2 = ((q2 + 64) & 127) - 64; const int d3 = (q2 - d2) >> 7;
    int8_t* p = e + (int64_t)(k >> 5) * 128 + (k & 31);
    p[0] = (int8_t)d0; p[32] = (int8_t)d1; p[64] = (int8_t)d2; p[96] = (int8_t)d3;
  }
  for (int sh = 1; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh, 64);
  if (lane == 0) norms[row] = acc;
  if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// D[i,j] = half(half(sqrt(d2))^2) for rows [rowA0, rowA0+M) x all N columns, atomicMax rowmax.  EA = encoded rows of the
// row block, EB = encoded rows of the whole set.  symmetric: only tiles on/above the diagonal are launched, mirrored on store.
__global__ __launch_bounds__(256, 2) void gram_i8_kernel(const int8_t* __restrict__ EA, const int8_t* __restrict__ EB,
                                                         const long long* __restrict__ nA, const long long* __restrict__ nB, int M, int N, int nkb,
                                                         int rowA0, hbits* __restrict__ D, unsigned* __restrict__ rowmax, int symmetric,
                                                         const int* __restrict__ flag) {
  if (*flag) return;     // some feature left [-1, 1]: the caller falls back to the fp64 kernel
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * GI_T * GI_PITCH];
  const int tiles_n = (N + GI_T - 1) / GI_T, tiles_m = (M + GI_T - 1) / GI_T;
  int tm, tn;
  if (symmetric) {
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
  const bool mirror = symmetric && tn > tm;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
  // staging: 16-byte chunk c of row r (+32): 8 chunks = the 128 bytes of one k block of one row
  const int sc = tid & 7, sr = tid >> 3;
  const int ar0 = min(tm * GI_T + sr, M - 1), ar1 = min(tm * GI_T + sr + 32, M - 1);
  const int br0 = min(tn * GI_T + sr, N - 1), br1 = min(tn * GI_T + sr + 32, N - 1);
  const uint4* a0 = reinterpret_cast<const uint4*>(EA + (int64_t)ar0 * nkb * 128) + sc;
  const uint4* a1 = reinterpret_cast<const uint4*>(EA + (int64_t)ar1 * nkb * 128) + sc;
  const uint4* b0 = reinterpret_cast<const uint4*>(EB + (int64_t)br0 * nkb * 128) + sc;
  const uint4* b1 = reinterpret_cast<const uint4*>(EB + (int64_t)br1 * nkb * 128) + sc;
  uint4 pa0, pa1, pb0, pb1;
#define SSG_GL(KB) { pa0 = a0[(KB) * 8]; pa1 = a1[(KB) * 8]; pb0 = b0[(KB) * 8]; pb1 = b1[(KB) * 8]; }
#define SSG_LS(BUF)                                                                                  \
  {                                                                                                  \
    unsigned char* As_ = lds + (BUF) * (2 * GI_T * GI_PITCH);                                        \
    unsigned char* Bs_ = As_ + GI_T * GI_PITCH;                                                      \
    *reinterpret_cast<uint4*>(As_ + sr * GI_PITCH + sc * 16) = pa0;                                  \
    *reinterpret_cast<uint4*>(As_ + (sr + 32) * GI_PITCH + sc * 16) = pa1;                           \
    *reinterpret_cast<uint4*>(Bs_ + sr * GI_PITCH + sc * 16) = pb0;                                  \
    *reinterpret_cast<uint4*>(Bs_ + (sr + 32) * GI_PITCH + sc * 16) = pb1;                           \
  }
  v16i acc[7];
#pragma unroll
  for (int w = 0; w < 7; w++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[w][r] = 0;

  SSG_GL(0)
  SSG_LS(0)
  __syncthreads();
  for (int kb = 0; kb < nkb; kb++) {
    { const int kn = min(kb + 1, nkb - 1); SSG_GL(kn) }      // next block's L2 latency hides under this block's 16 MFMAs
    const unsigned char* As = lds + (kb & 1) * (2 * GI_T * GI_PITCH) + (wm * 32 + l32) * GI_PITCH + h * 16;
    const unsigned char* Bs = lds + (kb & 1) * (2 * GI_T * GI_PITCH) + GI_T * GI_PITCH + (wn * 32 + l32) * GI_PITCH + h * 16;
    v4i a[4], b[4];
#pragma unroll
    for (int L = 0; L < 4; L++) { a[L] = *reinterpret_cast<const v4i*>(As + L * 32); b[L] = *reinterpret_cast<const v4i*>(Bs + L * 32); }
    // digit products grouped by weight a+b; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int La = 0; La < 4; La++)
#pragma unroll
      for (int Lb = 0; Lb < 4; Lb++) acc[La + Lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[La], b[Lb], acc[La + Lb], 0, 0, 0);
    SSG_LS((kb + 1) & 1)       // other stage: its readers finished before the previous barrier (redundant after the last block)
    __syncthreads();
  }
#undef SSG_GL
#undef SSG_LS

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31 -> B row (j), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> A row (i).
  const int gj = tn * GI_T + wn * 32 + l32;
  const bool jok = gj < N;
  const long long nj = jok ? nB[gj] : 0;
  unsigned cmax = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int li = tm * GI_T + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const bool ok = jok && li < M;
    unsigned dd = 0;
    if (ok) {
      long long dot = 0;
#pragma unroll
      for (int w = 6; w >= 0; w--) dot = dot * 128 + (long long)acc[w][r];
      long long d2i = nA[li] + nj - 2 * dot;           // exact squared distance in units of 2^-48
      if (d2i < 0 || rowA0 + li == gj) d2i = 0;         // cannot be negative; cdist(x, x) diagonal is exactly 0
      const double s = (double)d2i * 3.5527136788005009e-15;   // 2^-48, exact (d2i < 2^53)
      const hbits hh = d2h(sqrt(s));                    // cdist(...).astype(float16)   rerank.py:61
      dd = h_mul(hh, hh);                               // np.power(half, 2)            rerank.py:62
      D[(int64_t)li * N + gj] = (hbits)dd;
      if (mirror) D[(int64_t)gj * N + li] = (hbits)dd;
      cmax = cmax > dd ? cmax : dd;
    }
    unsigned red = dd;                                  // row maximum over the 32 columns of this half-wave
#pragma unroll
    for (int sh = 1; sh < 32; sh <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)red, sh, 64); red = red > o ? red : o; }
    if (l32 == 0 && li < M) atomicMax(&rowmax[li], red);
  }
  if (mirror) {   // mirrored rows are this tile's columns: one lane per column and half-wave
    const unsigned o = (unsigned)__shfl_xor((int)cmax, 32, 64);
    cmax = cmax > o ? cmax : o;
    if (h == 0 && jok) atomicMax(&rowmax[gj], cmax);
  }
}

}  // namespace ssg

using namespace ssg;

extern "C" size_t ssg_gram_i8_encoded_bytes(int n, int d) { return (size_t)n * (size_t)((d + 31) / 32) * 128; }

// Digits + exact squared norms of n rows; *flag |= 1 when a half-rounded feature lies outside [-1, 1].  E:
// ssg_gram_i8_encoded_bytes(n, d) bytes, norms: n int64.  The caller zeroes *flag once per matrix.
extern "C" int ssg_gram_i8_encode(const float* x, int n, int d, void* E, int64_t* norms, int32_t* flag, hipStream_t stream) {
  if (n <= 0 || d <= 0) { ssg_set_error("ssg_gram_i8_encode: empty input"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(gram_i8_encode_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, n, d, (d + 31) / 32, (int8_t*)E, (long long*)norms, flag);
  SSG_LAUNCH_CHECK("gram_i8_encode_kernel");
  return SSG_OK;
}

// Same contract as ssg_sqdist_self_f16 (rows [row0,row0+nrows) x N of the half original distance + row maxima) from the
// encoded features; does nothing when *flag != 0 (the caller then runs ssg_sqdist_self_f16).
extern "C" int ssg_sqdist_self_i8(const void* E, const int64_t* norms, int N, int d, int row0, int nrows, uint16_t* D, uint32_t* rowmax,
                                  const int32_t* flag, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > N || d <= 0) {
    ssg_set_error("ssg_sqdist_self_i8: bad shape N=%d d=%d row0=%d nrows=%d", N, d, row0, nrows);
    return SSG_ERR_INVALID;
  }
  const int nkb = (d + 31) / 32;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream, rowmax, nrows, 0u);
  const int symmetric = (row0 == 0 && nrows == N) ? 1 : 0;
  const int T = (N + GI_T - 1) / GI_T;
  const int64_t tiles = symmetric ? (int64_t)T * (T + 1) / 2 : (int64_t)((nrows + GI_T - 1) / GI_T) * T;
  if (tiles > 0x7fffffff) { ssg_set_error("ssg_sqdist_self_i8: too many tiles"); return SSG_ERR_INVALID; }
  const int8_t* e = (const int8_t*)E;
  hipLaunchKernelGGL(gram_i8_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, e + (int64_t)row0 * nkb * 128, e, (const long long*)norms + row0,
                     (const long long*)norms, nrows, N, nkb, row0, D, rowmax, symmetric, flag);
  SSG_LAUNCH_CHECK("gram_i8_kernel");
  return SSG_OK;
}
