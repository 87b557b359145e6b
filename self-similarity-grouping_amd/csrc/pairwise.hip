// pairwise.hip -- K3/K4: squared-L2 distance blocks on the FP64 matrix cores.
//
// Replaces scipy cdist under reid/rerank.py:36-40 (target x source term) and
// reid/rerank.py:33,61-62 (target x target "original distance").  The reference takes
// sqrt(sum (x-y)^2) in float64 and rounds to half; to reproduce those half values the
// Gram form  d2 = (|x|^2 + |y|^2) - 2 x.y  is evaluated in float64 on
// v_mfma_f64_16x16x4_f64 (MFMA-bound, 2*M*N*d flop).  Row norms are taken from the SAME
// instruction sequence (norm_kernel) so duplicate rows give d2 == 0 exactly, like cdist.
//
// Block tile 128x128, BK=16 (double-buffered LDS stages), 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles (16 f64
// accumulators x4 = 128 VGPRs).  Operands are converted f32 -> (half ->) f64 while being
// staged into LDS in [k][row] order with a 16-double pad per k-row: fragment reads
// (ds_read_b64, lane = row + 16*k) are bank-conflict free.
#include "ssg_common.h"
#include <algorithm>

namespace ssg {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LDR = 144;  // LDR: padded rows per k slice

template <bool ROUND16>
__device__ __forceinline__ double cvt_in(float x) {
  return ROUND16 ? (double)h2f(f2h(x)) : (double)x;
}

// block b runs on XCD b%8 (observed); give each XCD a contiguous run of tiles so that
// neighbouring tiles (same A row-panel) share that XCD's L2.  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int nx = 8, q = nwg / nx, r = nwg % nx, x = b % nx, s = b / nx;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// norms[i] = sum_k x_ik^2 with the accumulation order of gram_kernel (diagonal of the
// Gram block).  One wave per 16 rows.
template <bool ROUND16>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ X, int n, int d, double* __restrict__ norms) {
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = lane_id();
  const int r0 = wave * 16;
  if (r0 >= n) return;
  const int row = r0 + (lane & 15), ko = lane >> 4;
  const bool ok = row < n;
  const float* xr = X + (int64_t)row * d;
  v4d acc = {0., 0., 0., 0.};
  for (int k0 = 0; k0 < d; k0 += 4) {
    const int k = k0 + ko;
    const double x = (ok && k < d) ? cvt_in<ROUND16>(xr[k]) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
  }
  // C/D layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg
  const int c = lane & 15;
  if ((lane >> 4) == (c & 3) && ok) {
    const int r = c >> 2;
    norms[row] = r == 0 ? acc[0] : r == 1 ? acc[1] : r == 2 ? acc[2] : acc[3];
  }
}

// MODE 0 (self):  A = B-side of the same set, inputs rounded to half; writes
//                 D[i,j] = half(half(sqrt(d2))^2) and atomicMax rowmax[i].
// MODE 1 (cross): A = target f32, B = source f32; atomicMin rowmin[i] of half(sqrt(d2)^2).
template <int MODE>
__global__ __launch_bounds__(256, 2) void gram_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      const double* __restrict__ nA, const double* __restrict__ nB,
                                                      int M, int N, int d, int rowA0, hbits* __restrict__ D,
                                                      unsigned* __restrict__ rowred, int symmetric) {
  constexpr bool R16 = (MODE == 0);
  // double-buffered LDS stages of BK=16 k-slices: [stage][A|B][k][row(+pad)] doubles
  __shared__ double lds[2 * 2 * BK * LDR];
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tm, tn;
  if (MODE == 0 && (symmetric & 1)) {
    // D is symmetric (same products, same k order): with the whole matrix on one GPU only the
    // T(T+1)/2 tiles on or above the diagonal are launched (row-major over the triangle, so every
    // XCD gets an equal share) and the strictly-upper ones are mirrored on store.
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
  const bool mirror = MODE == 0 && (symmetric & 1) && tn > tm;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging map: one row per thread, 8 consecutive k (2 x float4) of the 16-wide k slice
  const int srow = tid & 127, shalf = tid >> 7;
  const int arow = tm * BM + srow, brow = tn * BN + srow;
  const float* ap = A + (int64_t)(arow < M ? arow : 0) * d;
  const float* bp = B + (int64_t)(brow < N ? brow : 0) * d;
  const bool aok = arow < M, bok = brow < N;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0., 0., 0., 0.};

  // named prefetch registers (arrays of float4 tend to end up in scratch)
  float4 pa0, pa1, pb0, pb1;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define SSG_GL(KT)                                                                         \
  {                                                                                        \
    const int k_ = (KT) * BK + shalf * 8;                                                  \
    const bool in0 = k_ < d, in1 = k_ + 4 < d;                                             \
    const float4 va0 = *reinterpret_cast<const float4*>(ap + (in0 ? k_ : 0));              \
    const float4 va1 = *reinterpret_cast<const float4*>(ap + (in1 ? k_ + 4 : 0));          \
    const float4 vb0 = *reinterpret_cast<const float4*>(bp + (in0 ? k_ : 0));              \
    const float4 vb1 = *reinterpret_cast<const float4*>(bp + (in1 ? k_ + 4 : 0));          \
    pa0 = (aok && in0) ? va0 : z4; pa1 = (aok && in1) ? va1 : z4;                          \
    pb0 = (bok && in0) ? vb0 : z4; pb1 = (bok && in1) ? vb1 : z4;                          \
  }
#define SSG_LS(BUF)                                                                        \
  {                                                                                        \
    double* As_ = lds + (BUF) * (2 * BK * LDR);                                            \
    double* Bs_ = As_ + BK * LDR;                                                          \
    const int k_ = shalf * 8;                                                              \
    As_[(k_ + 0) * LDR + srow] = cvt_in<R16>(pa0.x); As_[(k_ + 1) * LDR + srow] = cvt_in<R16>(pa0.y); \
    As_[(k_ + 2) * LDR + srow] = cvt_in<R16>(pa0.z); As_[(k_ + 3) * LDR + srow] = cvt_in<R16>(pa0.w); \
    As_[(k_ + 4) * LDR + srow] = cvt_in<R16>(pa1.x); As_[(k_ + 5) * LDR + srow] = cvt_in<R16>(pa1.y); \
    As_[(k_ + 6) * LDR + srow] = cvt_in<R16>(pa1.z); As_[(k_ + 7) * LDR + srow] = cvt_in<R16>(pa1.w); \
    Bs_[(k_ + 0) * LDR + srow] = cvt_in<R16>(pb0.x); Bs_[(k_ + 1) * LDR + srow] = cvt_in<R16>(pb0.y); \
    Bs_[(k_ + 2) * LDR + srow] = cvt_in<R16>(pb0.z); Bs_[(k_ + 3) * LDR + srow] = cvt_in<R16>(pb0.w); \
    Bs_[(k_ + 4) * LDR + srow] = cvt_in<R16>(pb1.x); Bs_[(k_ + 5) * LDR + srow] = cvt_in<R16>(pb1.y); \
    Bs_[(k_ + 6) * LDR + srow] = cvt_in<R16>(pb1.z); Bs_[(k_ + 7) * LDR + srow] = cvt_in<R16>(pb1.w); \
  }
  const int nk = (d + BK - 1) / BK;
  const int l16 = lane & 15, lk = lane >> 4;
  SSG_GL(0)
  SSG_LS(0)
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    if (kt + 1 < nk) SSG_GL(kt + 1)   // next slice's HBM/L2 latency hides under this slice's MFMAs
    const double* As = lds + (kt & 1) * (2 * BK * LDR);
    const double* Bs = As + BK * LDR;
    // fragments of k-step kk+1 are read from LDS while the 16 MFMAs of step kk run
    double a[2][4], b[2][4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[0][i] = As[lk * LDR + wm * 64 + i * 16 + l16];
#pragma unroll
    for (int j = 0; j < 4; j++) b[0][j] = Bs[lk * LDR + wn * 64 + j * 16 + l16];
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 4) {
        const int k = (kk + 1) * 4 + lk;
#pragma unroll
        for (int i = 0; i < 4; i++) a[nxt][i] = As[k * LDR + wm * 64 + i * 16 + l16];
#pragma unroll
        for (int j = 0; j < 4; j++) b[nxt][j] = Bs[k * LDR + wn * 64 + j * 16 + l16];
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) SSG_LS((kt + 1) & 1)   // other stage: its readers finished before the previous barrier
    __syncthreads();
  }
#undef SSG_GL
#undef SSG_LS

  // epilogue.  f64 MFMA C/D layout: col = lane&15, row = (lane>>4) + 4*reg.
  unsigned cmax[4] = {0u, 0u, 0u, 0u};   // mirror: running max of the column this lane owns in tile j
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int li = tm * BM + wm * 64 + i * 16 + lk + 4 * r;  // row within this call
      const bool rok = li < M;
      const double ni = rok ? nA[li] : 0.0;
      unsigned red = MODE == 0 ? 0u : 0xffffffffu;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int gj = tn * BN + wn * 64 + j * 16 + l16;
        if (rok && gj < N) {
          double s = (ni + nB[gj]) - 2.0 * acc[i][j][r];
          s = s > 0.0 ? s : 0.0;
          if (MODE == 0) {
            if (rowA0 + li == gj) s = 0.0;             // cdist(x, x) diagonal is exactly 0
            const double sq = sqrt(s);
            const hbits h = d2h(sq);                    // cdist(...).astype(float16)   rerank.py:61
            // np.power(half, 2) rerank.py:62; MemorySave branch (:49-59): np.power(cdist, 2).astype(float16), one rounding
            const hbits dd = (symmetric & 2) ? d2h(sq * sq) : h_mul(h, h);
            D[(int64_t)li * N + gj] = dd;
            red = red > dd ? red : (unsigned)dd;
            if (mirror) { D[(int64_t)gj * N + li] = dd; cmax[j] = cmax[j] > dd ? cmax[j] : (unsigned)dd; }
          } else {
            const double dist = sqrt(s);
            const hbits h = d2h(dist * dist);           // np.power(cdist, 2).astype(float16)  rerank.py:36-37
            red = red < h ? red : (unsigned)h;
          }
        }
      }
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)red, sh, 64);
        red = MODE == 0 ? (red > o ? red : o) : (red < o ? red : o);
      }
      if (l16 == 0 && rok) {
        if (MODE == 0) atomicMax(&rowred[li], red); else atomicMin(&rowred[li], red);
      }
    }
  }
  if (mirror) {
    // mirrored rows are this tile's columns: reduce over the 4 lanes (lk) that share a column
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned m = cmax[j];
      m = max(m, (unsigned)__shfl_xor((int)m, 16, 64));
      m = max(m, (unsigned)__shfl_xor((int)m, 32, 64));
      const int gj = tn * BN + wn * 64 + j * 16 + l16;
      if (lk == 0 && gj < N) atomicMax(&rowred[gj], m);
    }
  }
}

__global__ void fill_u32_kernel(unsigned* p, int n, unsigned v) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) p[i] = v;
}

// rerank.py:38-40 tail: v_raw = half(1 - exp(-minh)); v = v_raw / max(v_raw).
// (1-exp(-x) is monotone in x, so min_s f(x_s) == f(min_s x_s).)  Single block.
__global__ __launch_bounds__(1024) void source_vec_finish_kernel(const unsigned* __restrict__ rowmin, int N,
                                                                  hbits* __restrict__ v, unsigned* __restrict__ max_bits) {
  __shared__ unsigned smax[16];
  unsigned m = 0;
  for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
    const hbits o = h_sub(H_ONE, h_exp_neg((hbits)rowmin[i]));
    v[i] = o;
    m = m > o ? m : (unsigned)o;  // values are >= 0: bit order == value order
  }
  for (int sh = 1; sh < 64; sh <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, sh, 64); m = m > o ? m : o; }
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = m > smax[w] ? m : smax[w]; smax[0] = m; *max_bits = m; }
  __syncthreads();
  const hbits mx = (hbits)smax[0];
  for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) v[i] = h_div(v[i], mx);
}

}  // namespace ssg

namespace ssg {
// one wave per row: max |x| and the row's float32 norm, inflated so that it is an UPPER bound whatever order the float32 sum of
// squares is taken in: a lane sums d/64 squares sequentially (each rounded once), six cross-lane adds follow -- relative error of the
// sum <= (d/64 + 7) * 2^-24 * 1.01 -- and the square root adds half an ulp; the factor below covers twice that, with d (a fixed
// 1 + 1e-5 was only rigorous up to d ~ 32 k).  Non-negative floats order like their bit patterns, NaN patterns sort above +inf
__global__ __launch_bounds__(256) void range_stats_kernel(const float* __restrict__ a, int rows_a, const float* __restrict__ b, int rows_b, int d,
                                                          unsigned* __restrict__ out4) {
  // persistent waves over the rows; 16-byte loads when the rows allow it; one atomic per wave and statistic at the very end (a
  // per-row atomicMax on four words serialises in the L2: 140 k rows took 3.2 ms, the 1.15 GB read itself takes 0.25 ms)
  const int lane = lane_id();
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  unsigned best[4] = {0u, 0u, 0u, 0u};
  const bool vec = (d & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | (b ? reinterpret_cast<uintptr_t>(b) : 0)) & 15) == 0;
  const int rows = rows_a + rows_b;
  const float inflate = 1.0f + ((float)(d / 64 + 8) * 5.9604645e-8f + 1e-6f);       // (d/64 + 8) * 2^-24 on the sum >= its effect on the root; + 1e-6 slack
  // four rows per trip: their loads are all in flight before the first reduction (one row per trip left the wave waiting on its own
  // 8 KB and on twelve dependent cross-lane steps per row)
  for (int r0 = wave * 4; r0 < rows; r0 += nwaves * 4) {
    unsigned mx[4] = {0u, 0u, 0u, 0u}; float sq[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xp[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int row = min(r0 + u, rows - 1); xp[u] = row >= rows_a ? b + (int64_t)(row - rows_a) * d : a + (int64_t)row * d; }
    if (vec) {
      for (int c = lane * 4; c < d; c += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(xp[u] + c);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int t = 0; t < 4; t++) { const unsigned bits = __float_as_uint(e[t]) & 0x7fffffffu; mx[u] = bits > mx[u] ? bits : mx[u]; sq[u] += e[t] * e[t]; }
        }
      }
    } else {
      for (int c = lane; c < d; c += 64) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const float v = xp[u][c]; const unsigned bits = __float_as_uint(v) & 0x7fffffffu; mx[u] = bits > mx[u] ? bits : mx[u]; sq[u] += v * v; }
      }
    }
    for (int sh = 1; sh < 64; sh <<= 1) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const unsigned o = (unsigned)__shfl_xor((int)mx[u], sh, 64);
        mx[u] = o > mx[u] ? o : mx[u];
        sq[u] += __shfl_xor(sq[u], sh, 64);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (r0 + u >= rows) break;
      const bool second = r0 + u >= rows_a;
      const unsigned nb = __float_as_uint(sqrtf(sq[u]) * inflate) & 0x7fffffffu;
      unsigned& bm = best[second ? 1 : 0]; bm = mx[u] > bm ? mx[u] : bm;
      unsigned& bn = best[second ? 3 : 2]; bn = nb > bn ? nb : bn;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < 4; u++) if (best[u] > *reinterpret_cast<volatile unsigned*>(out4 + u)) atomicMax(out4 + u, best[u]);
  }
}
}  // namespace ssg

using namespace ssg;

// The value ranges the re-rank pipeline decides on (digit count of the exact Gram, operand scales and the rigorous tolerance of the
// source term's bound pass): out4 = [max|a|, max|b|, max row norm of a, max row norm of b] as float32 (norms are upper bounds), in ONE
// launch -- replaces five torch reductions per split (round 4).  b may be NULL (rows_b = 0).  NaN inputs give NaN outputs.
extern "C" int ssg_range_stats_f32(const float* a, int rows_a, const float* b, int rows_b, int d, float* out4, hipStream_t stream) {
  if (!a || rows_a <= 0 || d <= 0 || rows_b < 0 || (rows_b > 0 && !b) || !out4) { ssg_set_error("ssg_range_stats_f32: bad arguments"); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemsetAsync(out4, 0, 4 * sizeof(float), stream));
  const int rows = rows_a + rows_b;
  const int blocks = std::min((rows + 15) / 16, 2048);          // persistent waves (at most 32 per CU), four rows per trip
  hipLaunchKernelGGL(range_stats_kernel, dim3(blocks), dim3(256), 0, stream, a, rows_a, b, rows_b, d, reinterpret_cast<unsigned*>(out4));
  SSG_LAUNCH_CHECK("range_stats_kernel");
  return SSG_OK;
}

extern "C" int ssg_row_norms_f64(const float* x, int n, int d, int round_to_half, double* norms, hipStream_t stream) {
  if (n <= 0 || d <= 0) { ssg_set_error("ssg_row_norms_f64: empty input"); return SSG_ERR_INVALID; }
  const int waves = (n + 15) / 16, blocks = (waves + 3) / 4;
  if (round_to_half) hipLaunchKernelGGL(norm_kernel<true>, dim3(blocks), dim3(256), 0, stream, x, n, d, norms);
  else hipLaunchKernelGGL(norm_kernel<false>, dim3(blocks), dim3(256), 0, stream, x, n, d, norms);
  SSG_LAUNCH_CHECK("norm_kernel");
  return SSG_OK;
}

extern "C" int ssg_sqdist_self_f16(const float* x, const double* norms, int N, int d, int row0, int nrows, int memory_save, uint16_t* D,
                                   uint32_t* rowmax, hipStream_t stream) {
  if (N <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > N || (d & 3)) {
    ssg_set_error("ssg_sqdist_self_f16: bad shape N=%d d=%d row0=%d nrows=%d (d must be a multiple of 4)", N, d, row0, nrows);
    return SSG_ERR_INVALID;
  }
  hipLaunchKernelGGL(fill_u32_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream, rowmax, nrows, 0u);
  const int tiles = ((nrows + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int symmetric = (row0 == 0 && nrows == N) ? 1 : 0;   // whole matrix on this GPU: compute the upper triangle, mirror on store
  const int T = (N + BN - 1) / BN;
  hipLaunchKernelGGL(gram_kernel<0>, dim3(symmetric ? T * (T + 1) / 2 : tiles), dim3(256), 0, stream, x + (int64_t)row0 * d, x, norms + row0, norms, nrows, N, d,
                     row0, D, rowmax, symmetric | (memory_save ? 2 : 0));
  SSG_LAUNCH_CHECK("gram_kernel<self>");
  return SSG_OK;
}

extern "C" int ssg_source_rowmin_f16(const float* tgt, const double* ntgt, const float* src, const double* nsrc, int nrows, int Ns, int d,
                                     uint32_t* rowmin, hipStream_t stream) {
  if (nrows <= 0 || Ns <= 0 || (d & 3)) { ssg_set_error("ssg_source_rowmin_f16: bad shape"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(fill_u32_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream, rowmin, nrows, 0xffffffffu);
  const int tiles = ((nrows + BM - 1) / BM) * ((Ns + BN - 1) / BN);
  hipLaunchKernelGGL(gram_kernel<1>, dim3(tiles), dim3(256), 0, stream, tgt, src, ntgt, nsrc, nrows, Ns, d, 0, (uint16_t*)nullptr, rowmin, 0);
  SSG_LAUNCH_CHECK("gram_kernel<cross>");
  return SSG_OK;
}

extern "C" int ssg_source_vec_finish(const uint32_t* rowmin, int N, uint16_t* v, uint32_t* max_bits, hipStream_t stream) {
  if (N <= 0) { ssg_set_error("ssg_source_vec_finish: N<=0"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(source_vec_finish_kernel, dim3(1), dim3(1024), 0, stream, rowmin, N, v, max_bits);
  SSG_LAUNCH_CHECK("source_vec_finish_kernel");
  return SSG_OK;
}
