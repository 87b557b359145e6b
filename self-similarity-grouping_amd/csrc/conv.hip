// conv.hip -- K1/K2: ResNet-50 embedding forward (eval mode) on the matrix cores.
//
// Replaces the cuDNN path under reid/models/resnet.py:86-111 (torchvision ResNet-50 =
// reid/models/base.py:57-152 Bottleneck x [3,4,6,3]) as driven by
// reid/feature_extraction/cnn.py:10-22 and reid/evaluators.py:18-60.
//
// Every convolution is an implicit GEMM  out[m, n] = sum_k A[m, k] * W[n, k]  with
//   m = output pixel (b, oh, ow)   -- NHWC activations, so a pixel's channels are contiguous
//   n = output channel             -- weights stored [Cout][K], k = ((c/32)*KH*KW + r*KW + s)*32 + c%32:
//                                     channel chunks outermost, so that the taps of a chunk re-read cached pixels
// Eval-mode BatchNorm is folded into the weights/bias on the host; bias + residual add + ReLU are fused into the
// epilogue, so a bottleneck block costs 3 launches (the downsample branch rides in conv3's K) and no elementwise passes.
//
// Two element types share one kernel template (same 4 bytes per value, same addressing, same staging code):
//   SPLIT = true  (default path): fp32 values carried as hi+lo halves, x*w = xh*wh + xh*wl + xl*wh on
//                 v_mfma_f32_32x32x16_f16 with fp32 accumulation -- see "split-half activations" below;
//   SPLIT = false: plain fp32 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak = 1/16 of the fp16 rate).
//
// Tile: 128x256 (8 waves), 128x128 or 128x64 (4 waves) outputs per workgroup, each wave 64x64 (64x32) = 2x2 (2x1)
// MFMA tiles of 32x32; BK = 32 or 16 fp32-sized values per stage, double-buffered LDS, one barrier per K tile.  A and W
// tiles are staged through LDS as [row][BK + 4 floats] (pitch/16 B odd: ds_read_b128 fragment reads and ds_write_b128
// staging writes are both bank-conflict free).  The MFMA is issued as D = W_tile * A_tile^T, the k order inside a
// k-step is permuted identically for A and W (the reduction does not care).
#include "ssg_common.h"
#include <cstdlib>
#include <type_traits>

namespace ssg {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// ---- split-half activations ---------------------------------------------------------------------------
// The fp32 matrix cores run at 1/16 of the fp16 rate on gfx950, so the SPLIT kernels carry every fp32 value v
// as two halves  hi = half(v), lo = half(v - hi)  (hi + lo reproduces 22 significand bits of v; |v| < 65504,
// absolute floor 2^-25) and evaluate  x*w = xh*wh + xh*wl + xl*wh  on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: half x half products are exact in fp32, the dropped xl*wl term is below 2^-22 |x*w|, so the
// result is fp32-class (measured against fp64 next to the pure fp32 MFMA kernel in tests/test_gpu_parity.py)
// at 3/16 of the fp32 MFMA cost.  Memory format ("h8l8"): per 8 consecutive channels 32 bytes =
// [8 x half hi][8 x half lo] -- the same 4 bytes per value and the same addresses as the fp32 layout, so the
// staging code below is shared by both element types.
__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float unpack_lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float unpack_hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
// 4 values -> (hi pair of dwords, lo pair of dwords)
__device__ __forceinline__ void split_encode4(const float4 v, uint2& hi, uint2& lo) {
  const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
  hi = make_uint2(pack_h2(h0, h1), pack_h2(h2, h3));
  lo = make_uint2(pack_h2((_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1)), pack_h2((_Float16)(v.z - (float)h2), (_Float16)(v.w - (float)h3)));
}
// any of the four hi halves infinite or NaN (the value did not fit the split-half format)
__device__ __forceinline__ bool split_hi_nonfinite(const uint2 hi) {
  const unsigned a = hi.x & 0x7fff7fffu, b = hi.y & 0x7fff7fffu;
  return (a & 0xffffu) >= 0x7c00u || (a >> 16) >= 0x7c00u || (b & 0xffffu) >= 0x7c00u || (b >> 16) >= 0x7c00u;
}
__device__ __forceinline__ float4 split_decode4(const uint2 hi, const uint2 lo) {
  return make_float4(unpack_lo(hi.x) + unpack_lo(lo.x), unpack_hi(hi.x) + unpack_hi(lo.x), unpack_lo(hi.y) + unpack_lo(lo.y), unpack_hi(hi.y) + unpack_hi(lo.y));
}

struct ConvParams {
  const float* in; const float* w; const float* bias; const float* res; float* out;
  int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, relu;
  int M, Kpad;   // M = B*OH*OW, Kpad = weight row length (multiple of 32)
  int variant;   // tuning knob (SSG_CONV_VARIANT), 0 = default
  unsigned in_bytes;   // size of the input tensor (buffer-resource bound)
  // optional second 1x1 input concatenated along K (fused downsample branch): k-tiles >= nk1 read in2
  const float* in2; int H2, W2, Cin2, stride2, nk1; unsigned in2_bytes;
  // epilogue mode 1 (pairwise distance, reid/evaluators.py:63-85): out = rowterm[m] + bias[col] - 2*acc
  const float* rowterm; int epi;
  // epilogue mode 3 (distance filter): no matrix store; per output row the minimum of rowterm+bias-2*acc over each
  // 8-column granule goes to tilemin[m * tmin_ld + column / 8]
  float* tilemin; int tmin_ld;
  // split-half format (see "split-half activations" below): which tensors are encoded, and the factor
  // that undoes the operand scaling (weights / features are pre-scaled by a power of two)
  int out_split, res_split; float acc_scale;
  // per-output-channel factor applied to the accumulator before the bias (nullptr = none): the split-half path pre-scales
  // every weight ROW by its own power of two (folded checkpoints have per-channel BN scales spanning orders of magnitude)
  const float* cscale = nullptr;
  // set to 1 when a value that does not fit the split-half output format (|v| >= 65520 or NaN) is encoded (nullptr = no check)
  int* overflow = nullptr;
  // 1 = only the hi x hi product of the split-half operands (a plain fp16 GEMM: 1/3 of the matrix work, error 2^-10 |x||y|;
  // used by the source-term bound pass, whose tolerance covers it); 3 = the fp32-class three-product form
  int products = 3;
  unsigned long long* prof = nullptr;   // SSG_DMA_PROF builds (tools/micro/conv_prof.hip): 8 s_memtime stamps per workgroup of conv_dma_kernel
};
#ifdef SSG_DMA_PROF
#define SSG_DMA_STAMP(I_) { if (p.prof && threadIdx.x == 0) p.prof[(size_t)blockIdx.x * 8 + (I_)] = __builtin_readcyclecounter(); }
#else
#define SSG_DMA_STAMP(I_)
#endif

constexpr int CLD32 = 36;   // LDS row pitch in floats for BK=32 (BK=16 uses 20): pitch/4 odd -> conflict-free b128

__device__ __forceinline__ int conv_xcd_remap(int b, int nwg) {
  const int nx = 8, q = nwg / nx, r = nwg % nx, x = b % nx, s = b / nx;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// ---- split-half multiply: one k-step = 16 channels = two 32-byte [hi8|lo8] groups; half-wave h takes group 2*step+h
// (A and W alike).  Fragments of a step live in a SplitFrags; the pipeline in the kernel loads step s+1 while the
// MFMAs of step s run, and the first step of a tile right after the barrier that published it.
template <int MT, int NT> struct SplitFrags { v8h ah[MT], al[MT], bh[NT], bl[NT]; };

typedef float v4f __attribute__((ext_vector_type(4)));
// TAP4 (stem): a 16-byte unit is one filter tap of a 4-channel pixel, [4 x half hi][4 x half lo] ("h4l4"); the 8 hi
// halves of a fragment are the hi parts of two consecutive taps.
template <int MT, int NT, int CLD, bool TAP4>
__device__ __forceinline__ void split_load_frags(const float* Arow, const float* Brow, const int kofs, SplitFrags<MT, NT>& f) {
#pragma unroll
  for (int i = 0; i < MT; i++) {
    const float* q = Arow + i * 32 * CLD + kofs;
    if (TAP4) {
      const float4 t0 = *reinterpret_cast<const float4*>(q), t1 = *reinterpret_cast<const float4*>(q + 4);
      const v4f hi = {t0.x, t0.y, t1.x, t1.y}, lo = {t0.z, t0.w, t1.z, t1.w};
      f.ah[i] = __builtin_bit_cast(v8h, hi); f.al[i] = __builtin_bit_cast(v8h, lo);
    } else { f.ah[i] = *reinterpret_cast<const v8h*>(q); f.al[i] = *reinterpret_cast<const v8h*>(q + 4); }
  }
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const float* q = Brow + j * 32 * CLD + kofs;
    if (TAP4) {
      const float4 t0 = *reinterpret_cast<const float4*>(q), t1 = *reinterpret_cast<const float4*>(q + 4);
      const v4f hi = {t0.x, t0.y, t1.x, t1.y}, lo = {t0.z, t0.w, t1.z, t1.w};
      f.bh[j] = __builtin_bit_cast(v8h, hi); f.bl[j] = __builtin_bit_cast(v8h, lo);
    } else { f.bh[j] = *reinterpret_cast<const v8h*>(q); f.bl[j] = *reinterpret_cast<const v8h*>(q + 4); }
  }
}

// x*w = xh*wl + xl*wh + xh*wh; consecutive MFMAs hit different accumulators (each is revisited every MT*NT issues)
template <int MT, int NT>
__device__ __forceinline__ void split_mma_step(const SplitFrags<MT, NT>& f, v16f (&acc)[MT][NT]) {
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
}

// One K tile on the split path.  f0 already holds step 0.  `mid` (the next tile's LDS stores) is issued after the
// first step's MFMAs, when the matrix pipe has 12 x 32 cycles of work queued.
template <int MT, int NT, int CLD, int CBK, bool TAP4, typename F>
__device__ __forceinline__ void split_tile_mma(const float* Arow, const float* Brow, const int h, SplitFrags<MT, NT>& f0, SplitFrags<MT, NT>& f1,
                                               v16f (&acc)[MT][NT], F mid) {
  constexpr int KS = CBK / 16;
  if (KS == 2) split_load_frags<MT, NT, CLD, TAP4>(Arow, Brow, 16 + h * 8, f1);
  __builtin_amdgcn_sched_barrier(0);
  split_mma_step<MT, NT>(f0, acc);
  __builtin_amdgcn_sched_barrier(0);
  mid();
  __builtin_amdgcn_sched_barrier(0);
  if (KS == 2) split_mma_step<MT, NT>(f1, acc);
}

// One K tile of the fp32 multiply: Arow / Brow point at this lane's first row of the A (pixels) and W (channels) stage
// tiles; further MFMA tiles are 32 rows apart.  `mid` runs after the tile's MFMAs have been issued.
template <int MT, int NT, int CLD, int CBK, typename F>
__device__ __forceinline__ void conv_tile_mma(const float* Arow, const float* Brow, const int h, v16f (&acc)[MT][NT], F mid) {
    {
    // fragments for k-group g+1 are fetched from LDS while the MFMAs of group g run
    float4 a[2][MT], b[2][NT];
#pragma unroll
    for (int i = 0; i < MT; i++) a[0][i] = *reinterpret_cast<const float4*>(Arow + i * 32 * CLD + h * 4);
#pragma unroll
    for (int j = 0; j < NT; j++) b[0][j] = *reinterpret_cast<const float4*>(Brow + j * 32 * CLD + h * 4);
#pragma unroll
    for (int g = 0; g < CBK / 8; g++) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < CBK / 8) {
#pragma unroll
        for (int i = 0; i < MT; i++) a[nxt][i] = *reinterpret_cast<const float4*>(Arow + i * 32 * CLD + (g + 1) * 8 + h * 4);
#pragma unroll
        for (int j = 0; j < NT; j++) b[nxt][j] = *reinterpret_cast<const float4*>(Brow + j * 32 * CLD + (g + 1) * 8 + h * 4);
      }
      // consecutive MFMAs hit different accumulators (each one is revisited every MT*NT issues)
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][j].x, a[cur][i].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][j].y, a[cur][i].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][j].z, a[cur][i].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][j].w, a[cur][i].w, acc[i][j], 0, 0, 0);
      if (g == CBK / 8 - 1) mid();
    }
    }
}

// CIN4: stem mode, input has 4 channels (RGB + zero), one float4 = one filter tap.
// Workgroup tile BM x BN, 4 waves in a (BM/WM) x (BN/WN) grid, each wave WM x WN = MT x NT MFMA
// tiles of 32x32.  LDS is double buffered: the next K tile is fetched into registers while the
// current one is multiplied and written to the other buffer afterwards -> one barrier per K tile.
// The MFMA is issued as D = W_tile * A_tile^T (weights are the A operand), so a lane ends up
// with 4 consecutive output channels of one pixel per accumulator quad: bias/residual/output
// move as float4.
template <int BM, int BN, int WM, int WN, bool CIN4, int CBK, bool SPLIT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (BM / WM) * (BN / WN) == 4 ? 2 : 1) void conv_igemm_kernel(ConvParams p) {
  constexpr int WCOLS = BN / WN;
  constexpr int NWAVES = (BM / WM) * WCOLS, NTHREADS = NWAVES * 64;
  static_assert(NWAVES == 4 || NWAVES == 8, "4 or 8 waves per workgroup");
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int CLD = CBK + 4;              // LDS row pitch (floats)
  constexpr int KQ = CBK / 4;               // float4 per tile row
  constexpr int RPP = NTHREADS / KQ;        // tile rows staged per pass
  constexpr int AJ = BM / RPP, BJ = BN / RPP; // float4 loads per thread for the A / W tile
  constexpr int STAGE = (BM + BN) * CLD;    // floats per LDS stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // one declaration shared by the unity TU
  float* lds = reinterpret_cast<float*>(smem);

  const int tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = conv_xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int kq = tid & (KQ - 1), r0 = tid / KQ;

  // Staging state is kept in NAMED scalars, not arrays: hipcc left `float4 pb[BJ]` in scratch
  // memory (the prefetch then waited on every load to bounce it through the stack).
  static_assert((AJ == 4 || AJ == 2) && (BJ == 4 || BJ == 2 || BJ == 1), "staging code is written for 2 or 4 A rows and 1, 2 or 4 W rows per thread");
  int ab0, ab1, ab2 = -1, ab3 = -1, ah0, ah1, ah2 = 0, ah3 = 0, aw0, aw1, aw2 = 0, aw3 = 0;
#define SSG_ROW_INIT(J)                                                            \
  {                                                                                \
    const int m = tm * BM + r0 + RPP * J;                                          \
    if (m < p.M) {                                                                 \
      const int b = m / (p.OH * p.OW), rem = m - b * (p.OH * p.OW);                \
      const int oh = rem / p.OW, ow = rem - oh * p.OW;                             \
      ab##J = b; ah##J = oh * p.stride - p.pad; aw##J = ow * p.stride - p.pad;     \
    } else { ab##J = -1; ah##J = 0; aw##J = 0; }                                   \
  }
  SSG_ROW_INIT(0) SSG_ROW_INIT(1)
  if (AJ == 4) { SSG_ROW_INIT(2) SSG_ROW_INIT(3) }
#undef SSG_ROW_INIT
  const float* wbase = p.w + (int64_t)(tn * BN + r0) * p.Kpad + kq * 4;
  const int64_t wstep = (int64_t)RPP * p.Kpad;
  // two register sets: the SPLIT kernels finish a K tile five times faster than the fp32 ones, so the global
  // loads run TWO tiles ahead of the multiply (set kt&1 holds tile kt until it is written to LDS)
  constexpr bool PF2 = SPLIT;
  float4 pa0_0, pa1_0, pa2_0, pa3_0, pb0_0, pb1_0, pb2_0, pb3_0, pa0_1, pa1_1, pa2_1, pa3_1, pb0_1, pb1_1, pb2_1, pb3_1;
  pa2_0 = pa3_0 = pb1_0 = pb2_0 = pb3_0 = make_float4(0.f, 0.f, 0.f, 0.f);
  pa0_1 = pa1_1 = pa2_1 = pa3_1 = pb0_1 = pb1_1 = pb2_1 = pb3_1 = make_float4(0.f, 0.f, 0.f, 0.f);

  // A tile through a buffer resource: padding taps / rows beyond M get an out-of-range offset and
  // the hardware bounds check returns zeros -- no branch, no select (a load under a branch makes
  // hipcc wait for it on the spot and serialises the prefetch).
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t in2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2 ? p.in2 : p.in), 0, p.in2 ? p.in2_bytes : 0u, 0x00020000);
#define SSG_LOAD_A(J, S)                                                                                 \
  {                                                                                                     \
    const int ih = ah##J + r, iw = aw##J + s_;                                                          \
    const bool ok = ab##J >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;                           \
    const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1), bc = max(ab##J, 0);       \
    /* always-valid offset from clamped coordinates, then poisoned past the 2 GiB bound when !ok */     \
    const unsigned off = (unsigned)((((bc * p.H + ihc) * p.W + iwc) * p.Cin + c) * 4) + (ok ? 0u : 0x80000000u); \
    const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);                          \
    pa##J##_##S = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)); \
  }
// second input (fused downsample): 1x1, stride2, same output pixel grid
#define SSG_LOAD_A2(J, S)                                                                                 \
  {                                                                                                     \
    const int oh_ = (ah##J + p.pad) / p.stride, ow_ = (aw##J + p.pad) / p.stride;                       \
    const int bc = max(ab##J, 0);                                                                       \
    const unsigned off = (unsigned)((((bc * p.H2 + oh_ * p.stride2) * p.W2 + ow_ * p.stride2) * p.Cin2 + c2_) * 4) + (ab##J >= 0 ? 0u : 0x80000000u); \
    const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(in2_rsrc, off, 0, 0);                         \
    pa##J##_##S = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)); \
  }
#define SSG_GLOAD(KT, S)                                                                                 \
  {                                                                                                     \
    if ((KT) < p.nk1) {                                                                                 \
      int r, s_, c;                                                                                     \
      if (CIN4) { const int tap = (KT) * KQ + kq; r = tap / p.KW; s_ = tap - r * p.KW; c = 0; if (tap >= p.KH * p.KW) r = -100000; /* -> !ok */ } \
      else { /* k order = [32-channel chunk][tap][32]: the 9 taps of a chunk re-read the same few KB (L1/L2 hits) */ \
        const int ntap = p.KH * p.KW, kt32 = (CBK == 32) ? (KT) : ((KT) >> 1), half = (CBK == 32) ? 0 : ((KT) & 1) * 16; \
        const int chunk = kt32 / ntap, tap = kt32 - chunk * ntap; r = tap / p.KW; s_ = tap - r * p.KW; c = chunk * 32 + half + kq * 4; } \
      SSG_LOAD_A(0, S) SSG_LOAD_A(1, S) if (AJ == 4) { SSG_LOAD_A(2, S) SSG_LOAD_A(3, S) }              \
    } else {                                                                                            \
      const int c2_ = ((KT) - p.nk1) * CBK + kq * 4;                                                    \
      SSG_LOAD_A2(0, S) SSG_LOAD_A2(1, S) if (AJ == 4) { SSG_LOAD_A2(2, S) SSG_LOAD_A2(3, S) }          \
    }                                                                                                   \
    pb0_##S = *reinterpret_cast<const float4*>(wbase + (KT) * CBK);                                         \
    if (BJ >= 2) pb1_##S = *reinterpret_cast<const float4*>(wbase + wstep + (KT) * CBK);                    \
    if (BJ == 4) {                                                                                      \
      pb2_##S = *reinterpret_cast<const float4*>(wbase + 2 * wstep + (KT) * CBK);                           \
      pb3_##S = *reinterpret_cast<const float4*>(wbase + 3 * wstep + (KT) * CBK);                           \
    }                                                                                                   \
  }
#define SSG_LSTORE(BUF, S)                                                                                \
  {                                                                                                     \
    float* As_ = lds + (BUF) * STAGE;                                                                   \
    float* Bs_ = As_ + BM * CLD;                                                                        \
    *reinterpret_cast<float4*>(As_ + (r0 + 0) * CLD + kq * 4) = pa0_##S;                                    \
    *reinterpret_cast<float4*>(As_ + (r0 + RPP) * CLD + kq * 4) = pa1_##S;                                  \
    if (AJ == 4) {                                                                                      \
      *reinterpret_cast<float4*>(As_ + (r0 + 2 * RPP) * CLD + kq * 4) = pa2_##S;                            \
      *reinterpret_cast<float4*>(As_ + (r0 + 3 * RPP) * CLD + kq * 4) = pa3_##S;                            \
    }                                                                                                   \
    *reinterpret_cast<float4*>(Bs_ + (r0 + 0) * CLD + kq * 4) = pb0_##S;                                    \
    if (BJ >= 2) *reinterpret_cast<float4*>(Bs_ + (r0 + RPP) * CLD + kq * 4) = pb1_##S;                     \
    if (BJ == 4) {                                                                                      \
      *reinterpret_cast<float4*>(Bs_ + (r0 + 2 * RPP) * CLD + kq * 4) = pb2_##S;                            \
      *reinterpret_cast<float4*>(Bs_ + (r0 + 3 * RPP) * CLD + kq * 4) = pb3_##S;                            \
    }                                                                                                   \
  }

  v16f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / CBK;
  const int l32 = lane & 31, h = lane >> 5;
  // co-resident workgroups start in lockstep and would hit their non-MFMA phases together;
  // distinct static priorities de-phase them so one wave's MFMAs cover the other's staging.
  if (p.variant & 1) {
    const int pr = ((int)blockIdx.x >> 8) & 3;
    if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  // Branch-free pipeline: the tile index of a prefetch is clamped to the last tile (a harmless reload) and the
  // stage write after the last tile is redundant, so no load or register definition sits under a condition
  // (hipcc otherwise waits for the loads at the join).
  const float* Arow0 = lds + (wm * WM + l32) * CLD;
  const float* Brow0 = lds + BM * CLD + (wn * WN + l32) * CLD;
  SSG_GLOAD(0, 0)
  SSG_LSTORE(0, 0)
  if (PF2) { const int k1 = min(1, nk - 1); SSG_GLOAD(k1, 1) }
  __syncthreads();
  if constexpr (SPLIT) {
    SplitFrags<MT, NT> f0, f1;
    split_load_frags<MT, NT, CLD, CIN4>(Arow0, Brow0, h * 8, f0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {            // unrolled by two so that the register set of a tile is static
      { const int kn = min(kt + 2, nk - 1); SSG_GLOAD(kn, 0) }
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetch at the top: hipcc otherwise sinks the weight loads to just before their use
      split_tile_mma<MT, NT, CLD, CBK, CIN4>(Arow0, Brow0, h, f0, f1, acc, [&]() { SSG_LSTORE(1, 1) });
      __syncthreads();
      split_load_frags<MT, NT, CLD, CIN4>(Arow0 + STAGE, Brow0 + STAGE, h * 8, f0);   // in flight while the loads below are issued
      { const int kn = min(kt + 3, nk - 1); SSG_GLOAD(kn, 1) }
      __builtin_amdgcn_sched_barrier(0);
      split_tile_mma<MT, NT, CLD, CBK, CIN4>(Arow0 + STAGE, Brow0 + STAGE, h, f0, f1, acc, [&]() { SSG_LSTORE(0, 0) });
      __syncthreads();
      split_load_frags<MT, NT, CLD, CIN4>(Arow0, Brow0, h * 8, f0);
    }
    if (kt < nk) {                            // odd tile count: the last tile sits in stage 0
      split_tile_mma<MT, NT, CLD, CBK, CIN4>(Arow0, Brow0, h, f0, f1, acc, []() {});
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; kt++) {
      { const int kn = min(kt + 1, nk - 1); SSG_GLOAD(kn, 0) }   // HBM/L2 latency hides under this tile's MFMAs
      // stage write into the other buffer (its readers finished before the last barrier)
      conv_tile_mma<MT, NT, CLD, CBK>(Arow0 + (kt & 1) * STAGE, Brow0 + (kt & 1) * STAGE, h, acc, [&]() { SSG_LSTORE((kt + 1) & 1, 0) });
      __syncthreads();
    }
  }

#undef SSG_LOAD_A
#undef SSG_LOAD_A2
#undef SSG_GLOAD
#undef SSG_LSTORE
  if constexpr (SPLIT) {   // undo the power-of-two operand scaling (exact)
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] *= p.acc_scale;
  }
  // epilogue.  D = W * A^T: C/D layout col = lane&31 -> pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  // -> channel; accumulator quad q holds channels 8q + 4h + {0,1,2,3} of one pixel.
  if (p.epi == 3) {
    // distance filter epilogue: per output row, the minimum over every 8-column granule of this wave's tile
    // (lane = row; accumulator quad q of tile j holds columns j*32 + 8q + 4h + {0..3}; the two half-waves
    // h = 0/1 together cover the 8 columns of granule (j, q)).  tilemin[m][col/8], 8 floats per lane-row.
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int m = tm * BM + wm * WM + i * 32 + l32;
      const float rt = p.rowterm[m < p.M ? m : 0];
#pragma unroll
      for (int j = 0; j < NT; j++) {
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 bias = *reinterpret_cast<const float4*>(p.bias + tn * BN + wn * WN + j * 32 + 8 * q + 4 * h);
          float mn = fminf(fminf((rt + bias.x) - 2.f * acc[i][j][4 * q], (rt + bias.y) - 2.f * acc[i][j][4 * q + 1]),
                           fminf((rt + bias.z) - 2.f * acc[i][j][4 * q + 2], (rt + bias.w) - 2.f * acc[i][j][4 * q + 3]));
          g[q] = fminf(mn, __shfl_xor(mn, 32, 64));
        }
        if (h == 0 && m < p.M)
          *reinterpret_cast<float4*>(p.tilemin + (int64_t)m * p.tmin_ld + (tn * BN + wn * WN + j * 32) / 8) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
    return;
  }
  const float* __restrict__ resp = p.res;
  float* __restrict__ outp = p.out;
  if (p.epi == 0 && !(p.variant & 2)) {
    // Convolution epilogue, coalesced: the accumulators have one PIXEL per lane (row stride Cout*4 bytes), so direct
    // stores touch 64 cache lines per instruction.  Each wave instead turns its tile through a private LDS patch, 32
    // pixels x WN channels at a time, and leaves with lanes running along the channels: every residual load / output
    // store instruction covers whole WN*4-byte row segments (the stage buffers are free after the last barrier).
    constexpr int EP = WN + 4;                 // patch pitch (floats): EP/4 odd -> conflict-free b128 both ways
    constexpr int CPR = WN / 4, RPI = 64 / CPR, ITS = 32 / RPI;
    static_assert(NWAVES * 32 * EP <= 2 * (BM + BN) * CLD, "epilogue patch fits in the stage buffers");
    float* patch = lds + wave * (32 * EP);
    const int chunk = lane % CPR, prow = lane / CPR, odd = lane & 1;
    const int col = tn * BN + wn * WN + chunk * 4;
    const float4 bias = *reinterpret_cast<const float4*>(p.bias + col);
    const float4 cs = p.cscale ? *reinterpret_cast<const float4*>(p.cscale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int mbase = tm * BM + wm * WM + i * 32;
      float4 rr[ITS];
      if (resp) {
#pragma unroll
        for (int it = 0; it < ITS; it++) {
          const int m = mbase + it * RPI + prow;
          rr[it] = *reinterpret_cast<const float4*>(resp + (int64_t)(m < p.M ? m : 0) * p.Cout + col);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(patch + l32 * EP + j * 32 + 8 * q + 4 * h) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      // same wave wrote it: LDS operations of one wave complete in order
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        const int m = mbase + it * RPI + prow;
        float4 v = *reinterpret_cast<const float4*>(patch + (it * RPI + prow) * EP + chunk * 4);
        v.x = v.x * cs.x + bias.x; v.y = v.y * cs.y + bias.y; v.z = v.z * cs.z + bias.z; v.w = v.w * cs.w + bias.w;
        if (resp) {
          float4 r4 = rr[it];
          if (p.res_split) {
            // even lane holds hi0..7 of the 8-channel group, odd lane lo0..7; each needs hi and lo of ITS four channels
            const unsigned s0 = odd ? __float_as_uint(r4.x) : __float_as_uint(r4.z), s1 = odd ? __float_as_uint(r4.y) : __float_as_uint(r4.w);
            const unsigned g0 = lane_xor1(s0), g1 = lane_xor1(s1);
            r4 = odd ? split_decode4(make_uint2(g0, g1), make_uint2(__float_as_uint(r4.z), __float_as_uint(r4.w)))
                     : split_decode4(make_uint2(__float_as_uint(r4.x), __float_as_uint(r4.y)), make_uint2(g0, g1));
          }
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        if (p.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        if (p.out_split) {
          uint2 hp, lp;
          split_encode4(v, hp, lp);
          if (p.overflow && split_hi_nonfinite(hp)) *p.overflow = 1;
          const uint2 send = odd ? hp : lp;
          const uint2 recv = make_uint2(lane_xor1(send.x), lane_xor1(send.y));
          const uint4 st = odd ? make_uint4(recv.x, recv.y, lp.x, lp.y) : make_uint4(hp.x, hp.y, recv.x, recv.y);
#ifdef SSG_IGEMM_NT_STORE       // A/B knob: the register-staged kernel's split-half stores with the nt cache policy
          if (m < p.M) { const v4u sv_ = {st.x, st.y, st.z, st.w}; __builtin_nontemporal_store(sv_, reinterpret_cast<v4u*>(outp + (int64_t)m * p.Cout + col)); }
#else
          if (m < p.M) *reinterpret_cast<uint4*>(outp + (int64_t)m * p.Cout + col) = st;
#endif
        } else if (m < p.M) *reinterpret_cast<float4*>(outp + (int64_t)m * p.Cout + col) = v;
      }
    }
    return;
  }
  // All residual loads are issued first (one round trip), then bias/add/ReLU/stores: the stores
  // may alias the residual as far as the compiler knows, so an interleaved loop would serialise
  // 16 load->use->store round trips per wave.
  float4 rr[MT][NT][4];
  if (resp) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int m = tm * BM + wm * WM + i * 32 + l32;
      const int64_t mrow = (int64_t)(m < p.M ? m : 0) * p.Cout;
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float* g = resp + mrow + tn * BN + wn * WN + j * 32 + 8 * q;
          if (p.res_split) {   // this lane's channels 4h..4h+3 of the group: hi at +8h bytes, lo at +16+8h bytes
            const uint2 hi = *reinterpret_cast<const uint2*>(g + 2 * h), lo = *reinterpret_cast<const uint2*>(g + 4 + 2 * h);
            rr[i][j][q] = make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(lo.x), __uint_as_float(lo.y));
          } else rr[i][j][q] = *reinterpret_cast<const float4*>(g + 4 * h);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; i++) {
    const int m = tm * BM + wm * WM + i * 32 + l32;
#pragma unroll
    for (int j = 0; j < NT; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int col = tn * BN + wn * WN + j * 32 + 8 * q + 4 * h;
        const float4 bias = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 cs = p.cscale ? *reinterpret_cast<const float4*>(p.cscale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 v = make_float4(acc[i][j][4 * q] * cs.x + bias.x, acc[i][j][4 * q + 1] * cs.y + bias.y, acc[i][j][4 * q + 2] * cs.z + bias.z,
                               acc[i][j][4 * q + 3] * cs.w + bias.w);
        if (p.epi == 2) {       // cosine form 2 - 2<x,y> (reid/rerank.py:182)
          v = make_float4(2.f - 2.f * acc[i][j][4 * q], 2.f - 2.f * acc[i][j][4 * q + 1], 2.f - 2.f * acc[i][j][4 * q + 2], 2.f - 2.f * acc[i][j][4 * q + 3]);
        } else if (p.epi == 1) {
          const float rt = p.rowterm[m < p.M ? m : 0];
          v = make_float4((rt + bias.x) - 2.f * acc[i][j][4 * q], (rt + bias.y) - 2.f * acc[i][j][4 * q + 1], (rt + bias.z) - 2.f * acc[i][j][4 * q + 2],
                          (rt + bias.w) - 2.f * acc[i][j][4 * q + 3]);
        }
        if (resp) {
          float4 r4 = rr[i][j][q];
          if (p.res_split) r4 = split_decode4(make_uint2(__float_as_uint(r4.x), __float_as_uint(r4.y)), make_uint2(__float_as_uint(r4.z), __float_as_uint(r4.w)));
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        if (p.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        if (p.out_split) {
          // group of 8 channels = [hi0..7][lo0..7]; this lane has channels 4h..4h+3, its partner (lane^32) the other
          // four: half-wave 0 ends up storing the 16 hi bytes, half-wave 1 the 16 lo bytes.
          uint2 hp, lp;
          split_encode4(v, hp, lp);
          if (p.overflow && split_hi_nonfinite(hp)) *p.overflow = 1;
          const uint2 send = h ? hp : lp;
          const uint2 recv = make_uint2((unsigned)__shfl_xor((int)send.x, 32, 64), (unsigned)__shfl_xor((int)send.y, 32, 64));
          const uint4 st = h ? make_uint4(recv.x, recv.y, lp.x, lp.y) : make_uint4(hp.x, hp.y, recv.x, recv.y);
#ifdef SSG_IGEMM_NT_STORE       // A/B knob: the register-staged kernel's split-half stores with the nt cache policy
          if (m < p.M) { const v4u sv_ = {st.x, st.y, st.z, st.w}; __builtin_nontemporal_store(sv_, reinterpret_cast<v4u*>(outp + (int64_t)m * p.Cout + col)); }
#else
          if (m < p.M) *reinterpret_cast<uint4*>(outp + (int64_t)m * p.Cout + col) = st;
#endif
        } else if (m < p.M) *reinterpret_cast<float4*>(outp + (int64_t)m * p.Cout + col) = v;
      }
    }
  }
}

// ---- LDS-DMA variant of the split-half GEMM (128x256 tile, 8 waves) ----------------------------------------------------
// The register-staged kernel above runs ONE 8-wave workgroup per CU on its 128x256 tile: two BK=32 stages take 110 KB of
// LDS and the two prefetch register sets 48 VGPRs.  Here the tiles go global -> LDS directly (buffer_load_dwordx4 ... lds,
// 16 bytes per lane on gfx950): no staging registers, no ds_write, BK=16 stages of 24 KB in three separate __shared__ arrays
// (so that the compiler can see that the DMA into one stage does not alias the fragment reads of the other and keeps it
// in flight under the MFMAs) -> 72 KB of LDS and < 128 VGPRs: two workgroups, 16 waves per CU.
// The DMA places lane l of a wave at LDS offset 16*l, i.e. rows of 64 bytes without padding; bank conflicts of the
// fragment reads are avoided by XOR-swizzling the 16-byte chunk index with (row>>2)&3 -- on the global side (a lane loads
// logical chunk pc ^ g(row) into physical slot pc) and again when the fragments are read.
#define SSG_LDSP(ptr_) ((__attribute__((address_space(3))) void*)(ptr_))

// BM x BN = 128 x 256 (8 waves, two workgroups per CU), 128 x 128 (4 waves) or 256 x 256 (16 waves, ONE workgroup per CU with the same
// 16 waves: a third fewer global -> LDS bytes per MFMA than 128 x 256, which is what bounds these kernels).
// DUAL: a second 1x1 input is concatenated along K (fused downsample branch); a compile-time switch, because the main loop is bound
// by instruction issue (every branch, scalar division and v_readfirstlane per k-tile shows).
// NS: LDS stages (NS - 1 k-tiles in flight).  The 1x1 convolutions that stream their pixel operand from HBM (K = 512 ... 2048) sat at
// 0.4 of the matrix peak AND 0.4 of the HBM rate with two 32 KB tiles in flight per CU: bytes in flight / memory latency was the
// bound (64 KB / ~3 us = 21 GB/s per CU).  The 256 x 256 kernel (one workgroup per CU, 96 of 160 KB of LDS) takes a fourth stage.
// WM_ = 128 (256 x 256 tile only): 8 waves of 128 x 64 instead of 16 of 64 x 64 -- 12 fragment reads per 24 MFMAs instead of 8 per 12 (a
// quarter fewer LDS bytes per MFMA), half the waves at every barrier, 2 waves per SIMD with up to 256 VGPRs each.  TWOSET: the fragments
// of k-tile t + 1 are read into a second register set right after the barrier that publishes them, under the last MFMA group of tile t
// (source_bound_dma_kernel's schedule).
template <int BN, bool ONEPROD = false, int BM = 128, bool DUAL = false, int NS = 3, int WM_ = 64, bool TWOSET = false>
__global__ __launch_bounds__((BM / WM_) * (BN / 64) * 64, BM == 256 ? 1 : (BN == 256 ? 4 : 3)) void conv_dma_kernel(ConvParams p) {   // 4 (3) waves per SIMD: at most 128 (168) VGPRs
  static_assert(NS == 3 || NS == 4, "three or four stages");
  static_assert(WM_ == 64 || (WM_ == 128 && BM == 256 && BN == 256 && !ONEPROD && NS == 4), "128-row wave tiles: the 256 x 256 three-product kernel");
  static_assert(!TWOSET || WM_ == 128, "second fragment set: 8-wave kernel only");
  constexpr int WM = WM_, WN = 64, MT = WM / 32, NT = 2, CBK = 16;
  constexpr int WCOLS = BN / WN, NW = (BM / WM) * WCOLS;             // 8 waves (2 x 4) for 128 x 256, 4 waves (2 x 2) for 128 x 128, 16 (4 x 4) for 256 x 256
  constexpr int ABLK = BM / 16 / NW, WBLK = BN / 16 / NW;            // 16-row A / W blocks per wave and stage: 1 or 2
  constexpr int STAGE_BYTES = (BM + BN) * 64;                       // [A: BM rows x 64 B][W: BN rows x 64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char st0[STAGE_BYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char st1[STAGE_BYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char st2[STAGE_BYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char st3[NS == 4 ? STAGE_BYTES : 1024];
  const int tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = conv_xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: LDS-DMA destinations go to M0)
  SSG_DMA_STAMP(0)
  const int wm = wave / WCOLS, wn = wave % WCOLS, l32 = lane & 31, h = lane >> 5;

  // ---- DMA addressing: this wave fills A rows [16*ABLK*w, +16*ABLK) and W rows [32w, 32w+32) of every stage
  const int drow = lane >> 2, pc = lane & 3;                        // row inside a 16-row block, physical 16-byte slot
  const int lc4 = (pc ^ ((drow >> 2) & 3)) * 4;                     // logical chunk (in fp32-sized units) this lane fetches
  const int nkA = DUAL ? p.nk1 * 2 : 0x7fffffff;                     // p.nk1 counts 32-wide tiles on the dual path
  int ab0, ah0, aw0, ab1 = -1, ah1 = 0, aw1 = 0, abase0 = 0, abase1 = 0;     // abase: byte offset of tap (0, 0), channel lc4 of this lane's pixel (may be negative)
  unsigned a2b0 = 0x80000000u, a2b1 = 0x80000000u;
#define SSG_ROW(J)                                                                                                   \
  {                                                                                                                  \
    const int m = tm * BM + (wave * ABLK + J) * 16 + drow;                                                            \
    if (m < p.M) {                                                                                                   \
      const int b = m / (p.OH * p.OW), rem = m - b * (p.OH * p.OW);                                                  \
      const int oh = rem / p.OW, ow = rem - oh * p.OW;                                                               \
      ab##J = b; ah##J = oh * p.stride - p.pad; aw##J = ow * p.stride - p.pad;                                       \
      abase##J = (((b * p.H + ah##J) * p.W + aw##J) * p.Cin + lc4) * 4;                                               \
      if (DUAL) a2b##J = (unsigned)((((b * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.Cin2 + lc4) * 4);       \
    } else { ab##J = -1; ah##J = 0; aw##J = 0; }                                                                     \
  }
  SSG_ROW(0)
  if (ABLK == 2) SSG_ROW(1)
#undef SSG_ROW
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t in2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2 ? p.in2 : p.in), 0, p.in2 ? p.in2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (unsigned)((int64_t)p.Cout * p.Kpad * 4), 0x00020000);
  const unsigned wo0 = (unsigned)(((tn * BN + wave * (16 * WBLK) + drow) * p.Kpad + lc4) * 4), wo1 = wo0 + (unsigned)(16 * p.Kpad * 4);
  // DMA cursor: the next k-tile to fetch is dn = ((chunk dch) * KH*KW + tap (dr, ds)) * 2 + half; it only ever moves forward by one
  // (clamped at the last tile), so the tap decode is a few scalar increments instead of two integer divisions per tile and wave,
  // and a lane's address is its tap-(0,0) offset + one uniform delta (out-of-range taps: the offset is replaced, not clamped).
  const int nk = p.Kpad / CBK;
  int dn = 0, dr = 0, ds = 0, dch = 0;
#ifndef SSG_DMA_AUX_A
#define SSG_DMA_AUX_A 0        /* cache policy of the activation DMA (2 = nt): A/B knob, measured neutral */
#endif
#ifndef SSG_DMA_AUX_W
#define SSG_DMA_AUX_W 0
#endif
#define SSG_DMA_A(J, ST)                                                                                             \
  {                                                                                                                  \
    if (!DUAL || dn < nkA) {                                                                                          \
      const bool ok = ab##J >= 0 && (unsigned)(ah##J + dr) < (unsigned)p.H && (unsigned)(aw##J + ds) < (unsigned)p.W;  \
      const unsigned aoff = ok ? (unsigned)(abase##J + ddelta_) : 0x80000000u;                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, SSG_LDSP(ST + (wave * ABLK + J) * 1024), 16, aoff, 0, 0, SSG_DMA_AUX_A);   \
    } else {                                                                                                          \
      const unsigned aoff = a2b##J + (unsigned)((dn - nkA) * CBK * 4);                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in2_rsrc, SSG_LDSP(ST + (wave * ABLK + J) * 1024), 16, aoff, 0, 0, SSG_DMA_AUX_A);  \
    }                                                                                                                 \
  }
#define SSG_DMA_NEXT(ST)                                                                                             \
  {                                                                                                                  \
    const int ddelta_ = ((dr * p.W + ds) * p.Cin + dch * 32 + (dn & 1) * 16) * 4;                                     \
    SSG_DMA_A(0, ST) if (ABLK == 2) SSG_DMA_A(1, ST)                                                                  \
    const unsigned kb_ = (unsigned)(dn * CBK * 4);                                                                    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, SSG_LDSP(ST + BM * 64 + wave * (1024 * WBLK)), 16, wo0 + kb_, 0, 0, SSG_DMA_AUX_W);   \
    if (WBLK == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, SSG_LDSP(ST + BM * 64 + wave * 2048 + 1024), 16, wo1 + kb_, 0, 0, SSG_DMA_AUX_W); \
    {                                        /* branch-free advance; the tail re-fetches the last tile (harmless, keeps the vmcnt accounting uniform) */ \
      const int adv_ = dn < nk - 1 ? 1 : 0;                                                                          \
      dn += adv_;                                                                                                    \
      ds += adv_ & ~dn & 1;                                                                                          \
      const int w1_ = ds == p.KW ? 1 : 0; ds = w1_ ? 0 : ds; dr += w1_;                                               \
      const int w2_ = dr == p.KH ? 1 : 0; dr = w2_ ? 0 : dr; dch += w2_;                                              \
    }                                                                                                                \
  }

  // ---- fragment addressing: lane (row l32 of a 32-row MFMA tile, k half h); swizzle g = (row>>2)&3 depends on l32 only
  const int g = (l32 >> 2) & 3;
  const int offh = ((2 * h) ^ g) * 16, offl = ((2 * h + 1) ^ g) * 16;
  const int arow = (wm * WM + l32) * 64, brow = BM * 64 + (wn * WN + l32) * 64;
#define SSG_MMA(ST)                                                                                                  \
  {                                                                                                                  \
    v8h ah_[MT], al_[MT], bh_[NT], bl_[NT];                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                                 \
      ah_[i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offh); al_[i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offl); } \
    _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                                 \
      bh_[j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offh); bl_[j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offl); } \
    if constexpr (!ONEPROD) {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_[j], al_[i], acc[i][j], 0, 0, 0);                      \
      _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl_[j], ah_[i], acc[i][j], 0, 0, 0);                      \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_[j], ah_[i], acc[i][j], 0, 0, 0);                        \
  }

  v16f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // Three stages, loads two K tiles ahead.  `__syncthreads()` would make hipcc wait for EVERY outstanding DMA (vmcnt(0)),
  // i.e. for the tile it issued a moment ago; the explicit pair below waits only for the older tile (each tile is 3 or 4 DMA
  // instructions per wave) and then publishes it.  The compiler's own tracking of the DMA -> LDS-array dependencies stays in
  // force for the fragment reads (separate __shared__ arrays per stage).
  // (NS stages: NS - 1 tiles in flight, the wait leaves the NS - 2 newest tiles' DMAs outstanding)
  // lgkmcnt(0) belongs to the protocol: the barrier also tells the other waves "I am done READING the stage you overwrite next",
  // and hipcc sinks the (register-only) MFMAs of the previous tile -- and with them the lgkmcnt waits for its fragment reads --
  // below this asm statement.  Without the wait a wave passed the barrier with fragment reads still queued in the LDS pipeline,
  // and under load another wave's DMA overwrote the stage first: a few wrong 16-byte chunks (hi or lo halves of a later k-tile) in
  // about two output tiles per 1600 -- found in round 3 by the tile-shape equality test, present since the kernel was written.
#define SSG_PUBLISH()                                                                                              \
  { constexpr int OUTST = (ABLK + WBLK) * (NS - 2);                                                                \
    if (OUTST == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                  \
    else if (OUTST == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");                             \
    else if (OUTST == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");                             \
    else if (OUTST == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");                             \
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  static_assert((ABLK + WBLK) * (NS - 2) == 8 || (ABLK + WBLK) * (NS - 2) == 6 || (ABLK + WBLK) * (NS - 2) == 4 || (ABLK + WBLK) * (NS - 2) == 3 ||
                (ABLK + WBLK) * (NS - 2) == 2, "vmcnt literal for this tile shape");
  // BPOS > 0 (three-product kernels): the publishing barrier sits INSIDE a tile's multiply -- after its first (BPOS = 1) or second
  // (BPOS = 2) group of four MFMAs, when the hardware has consumed every fragment of the tile anyway, so lgkmcnt(0) costs nothing and the
  // remaining MFMAs of the tile are still queued behind the barrier: the matrix pipe has work while the next tile's fragment reads are
  // in flight (with the barrier in front of the whole multiply every wave of the CU read LDS at the same time and the pipes drained).
  // All NS stages are in flight ahead of the first tile; the DMA of tile t + NS goes into the stage of tile t right after the barrier.
#ifndef SSG_DMA_BPOS
#define SSG_DMA_BPOS 2
#endif
  constexpr int BPOS = ONEPROD ? 0 : SSG_DMA_BPOS;
  constexpr int TDMA = ABLK + WBLK;                                  // DMA instructions per wave and tile
  const int nfull = nk / NS * NS;
  if constexpr (BPOS == 0) {
  SSG_DMA_NEXT(st0)
  SSG_DMA_NEXT(st1)
  if constexpr (NS == 4) SSG_DMA_NEXT(st2)
  for (int kt = 0; kt < nfull; kt += NS) {   // tile t lives in stage array t % NS; no exits inside (they doubled the accumulators)
    if constexpr (NS == 3) {
      SSG_PUBLISH();                        // tile kt landed in st0 for every wave; everybody is done reading st2 (tile kt-1)
      SSG_DMA_NEXT(st2)
      __builtin_amdgcn_sched_barrier(0);    // keep the DMA issue ahead of the multiply
      SSG_MMA(st0)
      SSG_PUBLISH();
      SSG_DMA_NEXT(st0)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st1)
      SSG_PUBLISH();
      SSG_DMA_NEXT(st1)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st2)
    } else {
      SSG_PUBLISH();                        // tile kt landed in st0 for every wave; everybody is done reading st3 (tile kt-1)
      SSG_DMA_NEXT(st3)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st0)
      SSG_PUBLISH();
      SSG_DMA_NEXT(st0)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st1)
      SSG_PUBLISH();
      SSG_DMA_NEXT(st1)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st2)
      SSG_PUBLISH();
      SSG_DMA_NEXT(st2)
      __builtin_amdgcn_sched_barrier(0);
      SSG_MMA(st3)
    }
  }
  if (nk - nfull >= 1) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_MMA(st0) }
  if (nk - nfull >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_MMA(st1) }
  if constexpr (NS == 4) { if (nk - nfull == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_MMA(st2) } }
  } else if constexpr (TWOSET) {
    v8h fah[2][MT], fal[2][MT], fbh[2][NT], fbl[2][NT];     // [register set][tile]
#define SSG_READS2(ST, S)                                                                                            \
    { _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                               \
        fah[S][i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offh); fal[S][i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offl); } \
      _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                               \
        fbh[S][j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offh); fbl[S][j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offl); } }
#define SSG_GA(S) { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[S][j], fal[S][i], acc[i][j], 0, 0, 0); }
#define SSG_GB(S) { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbl[S][j], fah[S][i], acc[i][j], 0, 0, 0); }
#define SSG_GC(S) { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[S][j], fah[S][i], acc[i][j], 0, 0, 0); }
    // k-tile t (fragments in set S): the first two MFMA groups, publish tile t + 1 (and: everybody is done reading this stage -- its
    // fragment reads were waited for one step ago), refill this stage with tile t + NS, read tile t + 1's fragments into the other set,
    // the third group
#define SSG_STEP2(ST, STN, S)                                                                                        \
    { SSG_GA(S) SSG_GB(S)                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TDMA * (NS - 2)) : "memory");                  \
      SSG_DMA_NEXT(ST)                                                                                               \
      SSG_READS2(STN, 1 - (S))                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      SSG_GC(S) }
    SSG_DMA_NEXT(st0)
    SSG_DMA_NEXT(st1)
    SSG_DMA_NEXT(st2)
    SSG_DMA_NEXT(st3)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TDMA * (NS - 1)) : "memory");     // tile 0 landed for everybody
    SSG_READS2(st0, 0)
    for (int kt = 0; kt < nfull; kt += NS) {
      SSG_STEP2(st0, st1, 0)
      SSG_STEP2(st1, st2, 1)
      SSG_STEP2(st2, st3, 0)
      SSG_STEP2(st3, st0, 1)
    }
    // left-over tiles (nk % NS): set 0 holds the fragments of tile nfull (published by the last barrier of the loop, or by the one above)
    if (nk - nfull >= 1) { SSG_GA(0) SSG_GB(0) SSG_GC(0) }
    if (nk - nfull >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_READS2(st1, 0) SSG_GA(0) SSG_GB(0) SSG_GC(0) }
    if (nk - nfull == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_READS2(st2, 0) SSG_GA(0) SSG_GB(0) SSG_GC(0) }
#undef SSG_STEP2
#undef SSG_GA
#undef SSG_GB
#undef SSG_GC
#undef SSG_READS2
  } else {
    v8h fah[MT], fal[MT], fbh[NT], fbl[NT];
#define SSG_READS(ST)                                                                                                \
    { _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                               \
        fah[i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offh); fal[i] = *reinterpret_cast<const v8h*>(ST + arow + i * 2048 + offl); } \
      _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                               \
        fbh[j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offh); fbl[j] = *reinterpret_cast<const v8h*>(ST + brow + j * 2048 + offl); } }
#define SSG_G1 { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[j], fal[i], acc[i][j], 0, 0, 0); }
#define SSG_G2 { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbl[j], fah[i], acc[i][j], 0, 0, 0); }
#define SSG_G3 { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[j], fah[i], acc[i][j], 0, 0, 0); }
    // one tile: fragments, the first MFMA groups, publish (tile t + 1 landed for everybody, everybody done reading this stage),
    // refill this stage with tile t + NS, the rest of the multiply
#ifndef SSG_DMA_SETPRIO
#define SSG_DMA_SETPRIO 0      /* 1: s_setprio 1 around the MFMA groups (round 6 A/B, MI355X guide T5) */
#endif
#if SSG_DMA_SETPRIO
#define SSG_PRIO(P_) __builtin_amdgcn_s_setprio(P_);
#else
#define SSG_PRIO(P_)
#endif
#define SSG_STEP(ST)                                                                                                 \
    { SSG_READS(ST)                                                                                                  \
      SSG_PRIO(1)                                                                                                    \
      SSG_G1                                                                                                         \
      if constexpr (BPOS == 2) SSG_G2                                                                                \
      SSG_PRIO(0)                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TDMA * (NS - 2)) : "memory");                  \
      SSG_DMA_NEXT(ST)                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      SSG_PRIO(1)                                                                                                    \
      if constexpr (BPOS == 1) SSG_G2                                                                                \
      SSG_G3                                                                                                         \
      SSG_PRIO(0) }
#define SSG_LAST(ST) { SSG_READS(ST) SSG_G1 SSG_G2 SSG_G3 }
    SSG_DMA_NEXT(st0)
    SSG_DMA_NEXT(st1)
    SSG_DMA_NEXT(st2)
    if constexpr (NS == 4) SSG_DMA_NEXT(st3)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TDMA * (NS - 1)) : "memory");     // tile 0 landed for everybody
    SSG_DMA_STAMP(1)
    for (int kt = 0; kt < nfull; kt += NS) {
      SSG_STEP(st0)
      SSG_STEP(st1)
      SSG_STEP(st2)
      if constexpr (NS == 4) SSG_STEP(st3)
    }
    // left-over tiles (nk % NS): the last publish of the loop covered tile nfull; the refills of the last steps re-fetched the
    // clamped last tile into stages nobody reads again
    if (nk - nfull >= 1) SSG_LAST(st0)
    if (nk - nfull >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_LAST(st1) }
    if constexpr (NS == 4) { if (nk - nfull == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_LAST(st2) } }
#undef SSG_READS
#undef SSG_G1
#undef SSG_G2
#undef SSG_G3
#undef SSG_STEP
#undef SSG_LAST
  }
  SSG_DMA_STAMP(2)
  __syncthreads();                        // drains the (clamped, redundant) tail DMAs before the stages become epilogue patches
  SSG_DMA_STAMP(3)
#undef SSG_PUBLISH
#undef SSG_DMA_NEXT
#undef SSG_DMA_A
#undef SSG_MMA
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] *= p.acc_scale;

  if (p.epi == 3) {
    // distance filter epilogue (source-term bound pass): per output row the minimum over every 8-column granule, as in
    // conv_igemm_kernel (same accumulator layout)
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int m = tm * BM + wm * WM + i * 32 + l32;
      const float rt = p.rowterm[m < p.M ? m : 0];
#pragma unroll
      for (int j = 0; j < NT; j++) {
        float gq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 bias = *reinterpret_cast<const float4*>(p.bias + tn * BN + wn * WN + j * 32 + 8 * q + 4 * h);
          float mn = fminf(fminf((rt + bias.x) - 2.f * acc[i][j][4 * q], (rt + bias.y) - 2.f * acc[i][j][4 * q + 1]),
                           fminf((rt + bias.z) - 2.f * acc[i][j][4 * q + 2], (rt + bias.w) - 2.f * acc[i][j][4 * q + 3]));
          gq[q] = fminf(mn, __shfl_xor(mn, 32, 64));
        }
        if (h == 0 && m < p.M)
          *reinterpret_cast<float4*>(p.tilemin + (int64_t)m * p.tmin_ld + (tn * BN + wn * WN + j * 32) / 8) = make_float4(gq[0], gq[1], gq[2], gq[3]);
      }
    }
    return;
  }
  // ---- epilogue: per (i, j) a 32-pixel x 32-channel patch through LDS (waves 0-3 in st0, 4-7 in st1), then row segments
  constexpr int EP = 36, CPR = 8, RPI = 8, ITS = 4;
  constexpr int PPS = STAGE_BYTES / (32 * EP * 4) < 6 ? STAGE_BYTES / (32 * EP * 4) : 6;   // patches per stage array (4608 B each): 3, 5 or 6
  static_assert(PPS * 32 * EP * 4 <= STAGE_BYTES && 3 * PPS >= NW, "epilogue patches fit in the stage arrays");
  float* patch = reinterpret_cast<float*>(wave < PPS ? st0 : (wave < 2 * PPS ? st1 : st2)) + (wave % PPS) * (32 * EP);
  const int chunk = lane % CPR, prow = lane / CPR, odd = lane & 1;
  const float* __restrict__ resp = p.res;
  float* __restrict__ outp = p.out;
#ifndef SSG_DMA_FAST_EPI
#define SSG_DMA_FAST_EPI 1
#endif
  // cache-policy bits (gfx950: 1 = sc0, 2 = nt, 16 = sc1) of the straight-line epilogue's residual loads / output stores, with and without a residual.
  // A conv3 + residual launch streams its residual (last use) and its 0.5 GB output once: with `nt` they take no line of the L2 that the pixel
  // tile of its four column tiles and the weights go through (layer3: 0.326 -> 0.306 ms, layer4: 0.231 -> 0.219; profiles/r06_ab_nt_policy.txt).  Stores of the
  // launches WITHOUT a residual stay cached: their 0.13 GB outputs are the next launch's operand (nt there: conv3 + residual 0.307 -> 0.318).
#ifndef SSG_DMA_RES_AUX
#define SSG_DMA_RES_AUX 2
#endif
#ifndef SSG_DMA_OUT_AUX_RES
#define SSG_DMA_OUT_AUX_RES 2
#endif
#ifndef SSG_DMA_OUT_AUX
#define SSG_DMA_OUT_AUX 0
#endif
  // Round 6: the epilogue of the embedding's own launches (split-half in and out, ReLU, per-channel scales) as straight-line code.  The
  // general loop below tests five run-time switches per 16-byte piece (residual? encoded? ReLU? encode? range flag?): hipcc turns them
  // into ~100 scalar branches with a full s_waitcnt in front of the first use, the residual pieces of a patch are requested right before
  // they are needed and nothing of the next patch is in flight meanwhile -- four exposed HBM round trips per wave, with every wave of the CU
  // in them at the same time.  Here the switches are compile-time, the residual pieces (and the bias / scale vectors) of patch n + 1 are
  // requested BEFORE patch n is turned through LDS (two patches = 8 KB per wave in flight), the lane-pair exchanges are selects instead of
  // divergent branches, and the range flag is OR-accumulated and tested once.  Same operations on the same values in the same order:
  // bit-identical output (tests/test_gpu_parity.py::test_conv_fast_epilogue_matches_the_general_one).
  if (SSG_DMA_FAST_EPI && p.variant != 7 && p.out_split && p.relu && p.cscale != nullptr && (resp == nullptr || p.res_split) &&
      ((int64_t)p.M + BM) * p.Cout * 4 < (int64_t)0xffffffff) {      // (+ BM: the byte offsets of the rows behind M in the last tile must not wrap)
    constexpr int NP = MT * NT;
    unsigned ovf = 0u;
    // residual / output through buffer resources: a lane's address is one constant byte offset + a wave-uniform one, rows behind M are
    // out of range (loads return 0, stores are dropped) -- no 64-bit address registers, no row predicates
    const unsigned tbytes = (unsigned)((int64_t)p.M * p.Cout * 4), rowb = (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resp ? resp : outp), 0, resp ? tbytes : 0u, 0x00020000);
    const unsigned lane_off = (unsigned)prow * rowb + (unsigned)chunk * 16u;
    const unsigned ubase = (unsigned)(tm * BM + wm * WM) * rowb + (unsigned)(tn * BN + wn * WN) * 4u;
    // patch n = (j = n / MT, i = n % MT): one bias / scale vector pair per j
    auto poff = [&](const int n, const int it) { return ubase + (unsigned)((n % MT) * 32 + it * RPI) * rowb + (unsigned)(n / MT) * 128u + lane_off; };
    auto run = [&](auto RES_) {
      constexpr bool RES = decltype(RES_)::value;
      v4u rr[ITS];                                            // (ext-vector types: arrays of HIP's float4 struct stay in scratch memory)
      v4f bias = *reinterpret_cast<const v4f*>(p.bias + tn * BN + wn * WN + chunk * 4), cs = *reinterpret_cast<const v4f*>(p.cscale + tn * BN + wn * WN + chunk * 4);
      v4f bias_n = bias, cs_n = cs;
      if constexpr (RES) {
#pragma unroll
        for (int it = 0; it < ITS; it++) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, poff(0, it), 0, SSG_DMA_RES_AUX);
      }
#pragma unroll
      for (int n = 0; n < NP; n++) {
        const int j = n / MT, i = n % MT;
        if (n % MT == MT - 1 && j + 1 < NT) {                 // the next j's vectors, one patch ahead
          bias_n = *reinterpret_cast<const v4f*>(p.bias + tn * BN + wn * WN + (j + 1) * 32 + chunk * 4);
          cs_n = *reinterpret_cast<const v4f*>(p.cscale + tn * BN + wn * WN + (j + 1) * 32 + chunk * 4);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(patch + l32 * EP + 8 * q + 4 * h) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int it = 0; it < ITS; it++) {
          float4 v = *reinterpret_cast<const float4*>(patch + (it * RPI + prow) * EP + chunk * 4);
          v.x = v.x * cs[0] + bias[0]; v.y = v.y * cs[1] + bias[1]; v.z = v.z * cs[2] + bias[2]; v.w = v.w * cs[3] + bias[3];
          if constexpr (RES) {
            // even lane holds hi0..7 of the 8-channel group, odd lane lo0..7; each needs hi and lo of ITS four channels
            const unsigned a0 = rr[it][0], a1 = rr[it][1], a2 = rr[it][2], a3 = rr[it][3];
            if (n + 1 < NP) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, poff(n + 1, it), 0, SSG_DMA_RES_AUX);   // the next patch's piece, a whole patch ahead of its use
            const unsigned g0 = lane_xor1(odd ? a0 : a2), g1 = lane_xor1(odd ? a1 : a3);
            const float4 r4 = split_decode4(make_uint2(odd ? g0 : a0, odd ? g1 : a1), make_uint2(odd ? a2 : g0, odd ? a3 : g1));
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
          uint2 hp, lp;
          split_encode4(v, hp, lp);
          ovf |= ((hp.x & 0x7c007c00u) + 0x04000400u) | ((hp.y & 0x7c007c00u) + 0x04000400u);   // bits 15 / 31: a hi half with an all-ones exponent
          const unsigned rx = lane_xor1(odd ? hp.x : lp.x), ry = lane_xor1(odd ? hp.y : lp.y);
          const v4u stv = {odd ? rx : hp.x, odd ? ry : hp.y, odd ? lp.x : rx, odd ? lp.y : ry};
          __builtin_amdgcn_raw_buffer_store_b128(stv, out_rsrc, poff(n, it), 0, RES ? SSG_DMA_OUT_AUX_RES : SSG_DMA_OUT_AUX);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (n % MT == MT - 1) { bias = bias_n; cs = cs_n; }
      }
    };
    if (resp) run(std::true_type{}); else run(std::false_type{});
    if ((ovf & 0x80008000u) && p.overflow) *p.overflow = 1;
    SSG_DMA_STAMP(4)
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; i++) {
    const int mbase = tm * BM + wm * WM + i * 32;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int col = tn * BN + wn * WN + j * 32 + chunk * 4;
      const float4 bias = *reinterpret_cast<const float4*>(p.bias + col);
      const float4 cs = p.cscale ? *reinterpret_cast<const float4*>(p.cscale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
      float4 rr[ITS];
      if (resp) {
#pragma unroll
        for (int it = 0; it < ITS; it++) {
          const int m = mbase + it * RPI + prow;
          rr[it] = *reinterpret_cast<const float4*>(resp + (int64_t)(m < p.M ? m : 0) * p.Cout + col);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4*>(patch + l32 * EP + 8 * q + 4 * h) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        const int m = mbase + it * RPI + prow;
        float4 v = *reinterpret_cast<const float4*>(patch + (it * RPI + prow) * EP + chunk * 4);
        v.x = v.x * cs.x + bias.x; v.y = v.y * cs.y + bias.y; v.z = v.z * cs.z + bias.z; v.w = v.w * cs.w + bias.w;
        if (resp) {
          float4 r4 = rr[it];
          if (p.res_split) {
            const unsigned s0 = odd ? __float_as_uint(r4.x) : __float_as_uint(r4.z), s1 = odd ? __float_as_uint(r4.y) : __float_as_uint(r4.w);
            const unsigned g0 = lane_xor1(s0), g1 = lane_xor1(s1);
            r4 = odd ? split_decode4(make_uint2(g0, g1), make_uint2(__float_as_uint(r4.z), __float_as_uint(r4.w)))
                     : split_decode4(make_uint2(__float_as_uint(r4.x), __float_as_uint(r4.y)), make_uint2(g0, g1));
          }
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        if (p.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        if (p.out_split) {
          uint2 hp, lp;
          split_encode4(v, hp, lp);
          if (p.overflow && split_hi_nonfinite(hp)) *p.overflow = 1;
          const uint2 send = odd ? hp : lp;
          const uint2 recv = make_uint2(lane_xor1(send.x), lane_xor1(send.y));
          const uint4 stv = odd ? make_uint4(recv.x, recv.y, lp.x, lp.y) : make_uint4(hp.x, hp.y, recv.x, recv.y);
          if (m < p.M) *reinterpret_cast<uint4*>(outp + (int64_t)m * p.Cout + col) = stv;
        } else if (m < p.M) *reinterpret_cast<float4*>(outp + (int64_t)m * p.Cout + col) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// NCHW float32 images [B,3,H,W] -> NHWC4 [B,H,W,4] (4th channel 0), optional horizontal flip
// (reid/evaluators.py:12-16 fliplr fused into the layout change).
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int flip) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(x % W); const int64_t t = x / W; const int hh = (int)(t % H); const int b = (int)(t / H);
    const int ws = flip ? (W - 1 - w) : w;
    const int64_t plane = (int64_t)H * W, base = (int64_t)b * 3 * plane + (int64_t)hh * W + ws;
    reinterpret_cast<float4*>(out)[x] = make_float4(in[base], in[base + plane], in[base + 2 * plane], 0.f);
  }
}

// same, pixels written as split halves [h0 h1 h2 h3 | l0 l1 l2 l3] ("h4l4", 16 bytes per pixel)
__global__ void nchw_to_nhwc4_h4l4_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int flip) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(x % W); const int64_t t = x / W; const int hh = (int)(t % H); const int b = (int)(t / H);
    const int ws = flip ? (W - 1 - w) : w;
    const int64_t plane = (int64_t)H * W, base = (int64_t)b * 3 * plane + (int64_t)hh * W + ws;
    uint2 hi, lo;
    split_encode4(make_float4(in[base], in[base + plane], in[base + 2 * plane], 0.f), hi, lo);
    reinterpret_cast<uint4*>(out)[x] = make_uint4(hi.x, hi.y, lo.x, lo.y);
  }
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (reid/models/base.py:105); C % 4 == 0
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * OH * OW * C4;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(x % C4); int64_t t = x / C4;
    const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int b = (int)(t / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int r = 0; r < 3; r++) {
      const int ih = oh * 2 - 1 + r; if (ih < 0 || ih >= H) continue;
      for (int s = 0; s < 3; s++) {
        const int iw = ow * 2 - 1 + s; if (iw < 0 || iw >= W) continue;
        const float4 v = reinterpret_cast<const float4*>(in + ((int64_t)(b * H + ih) * W + iw) * C)[c4];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    reinterpret_cast<float4*>(out)[x] = m;
  }
}

// Global + stripe average pooling (reid/models/resnet.py:93-111): feature map [B,H,W,C] ->
// out[s][b][c], s = 0: whole map, s = 1..S: rows [H/S*(s-1), H/S*s).  S = 0/1 -> whole only.
__global__ void gap_stripes_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int S) {
  const int nsets = S > 1 ? S + 1 : 1;
  const int64_t total = (int64_t)nsets * B * C;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(x % C); int64_t t = x / C; const int b = (int)(t % B); const int s = (int)(t / B);
    int h0 = 0, h1 = H;
    if (s > 0) { const int hs = H / S; h0 = hs * (s - 1); h1 = hs * s; }
    float acc = 0.f;
    for (int hh = h0; hh < h1; hh++)
      for (int w = 0; w < W; w++) acc += in[((int64_t)(b * H + hh) * W + w) * C + c];
    out[x] = acc / (float)((h1 - h0) * W);
  }
}

// ---- split-half (h8l8) versions of the non-GEMM layers
__device__ __forceinline__ void h8l8_load8(const float* g, float4& a, float4& b) {
  const uint4 hi = *reinterpret_cast<const uint4*>(g), lo = *reinterpret_cast<const uint4*>(g + 4);
  a = split_decode4(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y));
  b = split_decode4(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w));
}
__device__ __forceinline__ void h8l8_store8(float* g, const float4 a, const float4 b) {
  uint2 ha, la, hb, lb;
  split_encode4(a, ha, la); split_encode4(b, hb, lb);
  *reinterpret_cast<uint4*>(g) = make_uint4(ha.x, ha.y, hb.x, hb.y);
  *reinterpret_cast<uint4*>(g + 4) = make_uint4(la.x, la.y, lb.x, lb.y);
}

__global__ void maxpool3x3s2_h8l8_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int OH, int OW) {
  const int C8 = C / 8;
  const int64_t total = (int64_t)B * OH * OW * C8;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(x % C8); int64_t t = x / C8;
    const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int b = (int)(t / OH);
    float4 m0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), m1 = m0;
    for (int r = 0; r < 3; r++) {
      const int ih = oh * 2 - 1 + r; if (ih < 0 || ih >= H) continue;
      for (int s = 0; s < 3; s++) {
        const int iw = ow * 2 - 1 + s; if (iw < 0 || iw >= W) continue;
        float4 a, c;
        h8l8_load8(in + ((int64_t)(b * H + ih) * W + iw) * C + c8 * 8, a, c);
        m0.x = fmaxf(m0.x, a.x); m0.y = fmaxf(m0.y, a.y); m0.z = fmaxf(m0.z, a.z); m0.w = fmaxf(m0.w, a.w);
        m1.x = fmaxf(m1.x, c.x); m1.y = fmaxf(m1.y, c.y); m1.z = fmaxf(m1.z, c.z); m1.w = fmaxf(m1.w, c.w);
      }
    }
    h8l8_store8(out + x * 8, m0, m1);
  }
}

__global__ void gap_stripes_h8l8_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int S) {
  const int nsets = S > 1 ? S + 1 : 1;
  const int64_t total = (int64_t)nsets * B * C;
  const unsigned short* hp = reinterpret_cast<const unsigned short*>(in);
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(x % C); int64_t t = x / C; const int b = (int)(t % B); const int s = (int)(t / B);
    int h0 = 0, h1 = H;
    if (s > 0) { const int hs = H / S; h0 = hs * (s - 1); h1 = hs * s; }
    float acc = 0.f;
    for (int hh = h0; hh < h1; hh++)
      for (int w = 0; w < W; w++) {
        const int64_t e = (((int64_t)(b * H + hh) * W + w) * C + (c & ~7)) * 2 + (c & 7);   // half index of the hi part
        acc += (float)__builtin_bit_cast(_Float16, hp[e]) + (float)__builtin_bit_cast(_Float16, hp[e + 8]);
      }
    out[x] = acc / (float)((h1 - h0) * W);
  }
}

// fp32 [n] (n % 8 == 0) <-> h8l8, with a power-of-two scale applied on the way in (out = split(in * scale))
__global__ void h8l8_encode_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t ngroups, float scale) {
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < ngroups; x += (int64_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(in)[2 * x], b = reinterpret_cast<const float4*>(in)[2 * x + 1];
    a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale; b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
    h8l8_store8(out + x * 8, a, b);
  }
}
__global__ void h8l8_decode_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t ngroups, float scale) {
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < ngroups; x += (int64_t)gridDim.x * blockDim.x) {
    float4 a, b;
    h8l8_load8(in + x * 8, a, b);
    a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale; b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
    reinterpret_cast<float4*>(out)[2 * x] = a; reinterpret_cast<float4*>(out)[2 * x + 1] = b;
  }
}

// out = (a + b) / ||a + b||_2 per row (reid/evaluators.py:31-35); one wave per row
__global__ __launch_bounds__(256) void flip_sum_l2norm_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                              int rows, int C) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= rows) return;
  const int lane = lane_id();
  const float* pa = a + (int64_t)row * C; const float* pb = b + (int64_t)row * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) { const float s = pa[c] + pb[c]; ss += s * s; }
  for (int sh = 1; sh < 64; sh <<= 1) ss += __shfl_xor(ss, sh, 64);
  const float nrm = sqrtf(ss);
  for (int c = lane; c < C; c += 64) out[(int64_t)row * C + c] = (pa[c] + pb[c]) / nrm;
}

}  // namespace ssg

using namespace ssg;

template <int BM, int BN, int WM, int WN, bool CIN4, int CBK, bool SPLIT>
static int launch_conv_bk(const ConvParams& p, hipStream_t stream) {
  const size_t lds = 2 * (size_t)(BM + BN) * (CBK + 4) * sizeof(float);
  const int tiles = ((p.M + BM - 1) / BM) * (p.Cout / BN);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    int rc = ssg_check_hip(hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, CIN4, CBK, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(conv)");
    if (rc) return rc;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, CIN4, CBK, SPLIT>), dim3(tiles), dim3((BM / WM) * (BN / WN) * 64), lds, stream, p);
  return ssg_check_hip(hipGetLastError(), "conv_igemm_kernel");
}
// BK=16 stages (default): half the LDS of BK=32 -> 3 instead of 2 co-resident workgroups per CU, so one
// workgroup's prologue / epilogue overlaps the others' MFMA phases.  The fused dual-input GEMM stays
// on BK=32.  SSG_CONV_BK16_MAXK=<K> restricts BK=16 to reductions of at most K (tuning knob).
template <int BM, int BN, int WM, int WN, bool CIN4>
static int launch_conv(const ConvParams& p, hipStream_t stream, bool split = false) {
  static int bk16_max = -1;   // measured on MI355X: BK=16 wins for every ResNet-50 layer (9.1k -> 9.8k img/s)
  if (bk16_max < 0) { const char* e = getenv("SSG_CONV_BK16_MAXK"); bk16_max = e ? atoi(e) : 0x7fffffff; }
  if (split) {
    if constexpr (CIN4) return launch_conv_bk<BM, BN, WM, WN, true, 16, true>(p, stream);   // one k-step = 4 taps
    else {
      static int sbk16 = -1;   // SSG_SPLIT_BK16=<K>: reductions of at most K use BK=16 stages
      if (sbk16 < 0) { const char* e = getenv("SSG_SPLIT_BK16"); sbk16 = e ? atoi(e) : 0; }     // round 6: 0 (was 256) -- short reductions take the LDS-DMA kernel below too
      if (!p.in2 && p.Kpad <= sbk16) return launch_conv_bk<BM, BN, WM, WN, false, 16, true>(p, stream);
      if constexpr (BM == 128 && BN == 128) {   // long reductions on 128x128 tiles: LDS-DMA kernel (SSG_CONV_DMA bit 1)
        static int dma = -1;
        // bit 1: round 2 measured it neutral (21.42 vs 21.46 k img/s); with the round-6 epilogue the layer2 entry launches gain (256 -> 128 1x1:
        // 0.723 -> 0.653 ms, 128 -> 128 3x3 stride 2: 0.591 -> 0.523 ms per 1000 images; bit-identical, profiles/r06_ab_nt_policy.txt): on by default
        if (dma < 0) { const char* e = getenv("SSG_CONV_DMA"); dma = e ? atoi(e) : 3; }
        if ((dma & 2) && p.epi == 0 && !p.in2) {
          const int tiles = ((p.M + 127) / 128) * (p.Cout / 128);
          hipLaunchKernelGGL(conv_dma_kernel<128>, dim3(tiles), dim3(256), 0, stream, p);
          return ssg_check_hip(hipGetLastError(), "conv_dma_kernel<128>");
        }
      }
      return launch_conv_bk<BM, BN, WM, WN, false, 32, true>(p, stream);
    }
  }
  if (!p.in2 && p.Kpad <= bk16_max) return launch_conv_bk<BM, BN, WM, WN, CIN4, 16, false>(p, stream);
  return launch_conv_bk<BM, BN, WM, WN, CIN4, 32, false>(p, stream);
}

// Few 128x128 tiles (deep layers: M = B*8*4) leave one workgroup per CU with nothing to overlap its staging
// with; 128x64 tiles double the workgroup count.  SSG_SPLIT_BN64_MAXTILES=<T>: use them below T tiles.
// 128x256 tiles (8 waves) read the pixel tile once for 256 output channels: 3/4 of the global->LDS bytes per flop of
// the 128x128 tile, which is what bounds the split kernels.  SSG_SPLIT_WIDE_MINTILES=<T>: use them when at least T
// such tiles exist (0 = never).
static bool conv_prefers_wide(const ConvParams& p, bool split) {
  static int mintiles = -1;
  if (mintiles < 0) { const char* e = getenv("SSG_SPLIT_WIDE_MINTILES"); mintiles = e ? atoi(e) : 256; }
  if (!split || mintiles <= 0 || (p.Cout % 256) || p.Kpad < 128) return false;   // short reductions: more, smaller workgroups overlap better
  return ((p.M + 127) / 128) * (p.Cout / 256) >= mintiles;
}

// the LDS-DMA kernel takes the convolution-epilogue 128x256 launches (SSG_CONV_DMA: bit 0 = 128x256, bit 1 = 128x128
// launches; 0 = register-staged kernels everywhere)
static int launch_conv_wide(const ConvParams& p, hipStream_t stream) {
  static int dma = -1;
  if (dma < 0) { const char* e = getenv("SSG_CONV_DMA"); dma = e ? atoi(e) : 3; }
  if ((dma & 1) && (p.epi == 0 || p.epi == 3) && (int64_t)p.Cout * p.Kpad * 4 < 0x7fffffffLL) {   // (weights go through a 2 GiB buffer resource)
    // 256 x 256 tiles (16 waves, one workgroup per CU): a third fewer global -> LDS bytes per MFMA; taken when they fill the chip
    // (SSG_CONV_TALL_MINTILES = least number of such tiles, 0 = never)
    static int tall = -1;
    if (tall < 0) { const char* e = getenv("SSG_CONV_TALL_MINTILES"); tall = e ? atoi(e) : 200; }
    const int tiles_tall = ((p.M + 255) / 256) * (p.Cout / 256);
    static int tall_dual = -1;               // SSG_CONV_TALL_DUAL=1: also for the fused conv3 | downsample GEMMs (tuning knob)
    if (tall_dual < 0) { const char* e = getenv("SSG_CONV_TALL_DUAL"); tall_dual = e ? atoi(e) : 0; }
    if (tall > 0 && p.products == 3 && p.epi == 0 && p.in2 && tall_dual && tiles_tall >= tall) {
      hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, true, 4>), dim3(tiles_tall), dim3(1024), 0, stream, p);
      return ssg_check_hip(hipGetLastError(), "conv_dma_kernel<256x256, dual>");
    }
    static int tall_res = -1;                // SSG_CONV_TALL_RES=1: also for convolutions with a residual epilogue (tuning knob)
    if (tall_res < 0) { const char* e = getenv("SSG_CONV_TALL_RES"); tall_res = e ? atoi(e) : 0; }
    if (tall > 0 && p.products == 3 && p.epi == 0 && (!p.res || tall_res) && !p.in2 && tiles_tall >= tall) {   // measured: -2.5 % (3x3) / -6 % (1x1) on layer3 shapes; with a residual epilogue +3 %
      static int tall_ns = -1;               // SSG_CONV_TALL_STAGES=3: the three-stage pipeline of round 2 (tuning knob)
      if (tall_ns < 0) { const char* e = getenv("SSG_CONV_TALL_STAGES"); tall_ns = e ? atoi(e) : 4; }
      static int tall_wm = -1;               // SSG_CONV_TALL_WM=128: 8 waves of 128 x 64; =129: the same with a second fragment register set
      if (tall_wm < 0) { const char* e = getenv("SSG_CONV_TALL_WM"); tall_wm = e ? atoi(e) : 64; }
      if (tall_wm == 128) hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4, 128, false>), dim3(tiles_tall), dim3(512), 0, stream, p);
      else if (tall_wm == 129) hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4, 128, true>), dim3(tiles_tall), dim3(512), 0, stream, p);
      else if (tall_ns == 4) hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4>), dim3(tiles_tall), dim3(1024), 0, stream, p);
      else hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false>), dim3(tiles_tall), dim3(1024), 0, stream, p);
      return ssg_check_hip(hipGetLastError(), "conv_dma_kernel<256x256>");
    }
    const int tiles = ((p.M + 127) / 128) * (p.Cout / 256);
    if (p.products == 1) hipLaunchKernelGGL((conv_dma_kernel<256, true>), dim3(tiles), dim3(512), 0, stream, p);
    else if (p.in2) hipLaunchKernelGGL((conv_dma_kernel<256, false, 128, true>), dim3(tiles), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv_dma_kernel<256>, dim3(tiles), dim3(512), 0, stream, p);
    return ssg_check_hip(hipGetLastError(), "conv_dma_kernel");
  }
  return launch_conv_bk<128, 256, 64, 64, false, 32, true>(p, stream);
}

static bool conv_prefers_bn64(const ConvParams& p, bool split) {
  static int maxtiles = -1;
  if (maxtiles < 0) { const char* e = getenv("SSG_SPLIT_BN64_MAXTILES"); maxtiles = e ? atoi(e) : 300; }
  if (!split) return false;
  const int tiles = ((p.M + 127) / 128) * (p.Cout / 128);
  return tiles < maxtiles;
}

// Conv2d(bias folded from eval BatchNorm) + optional residual add + optional ReLU, NHWC fp32.
//   in  [B,H,W,Cin]; w [Cout][Kpad] with k = ((c/32)*KH*KW + r*KW + s)*32 + c%32, rows zero-padded to Kpad
//   (multiple of 32); bias [Cout]; res/out [B,OH,OW,Cout].  Cin % 32 == 0, or Cin == 4 (stem:
//   RGB0 pixels, Kpad = 32*ceil(KH*KW/8)).  Cout % 64 == 0.
//
// flags: SSG_CONV_IN_SPLIT (1) = in and w are in the split-half h8l8 format (w additionally pre-multiplied by
// 1/acc_scale, a power of two), the GEMM runs on the fp16 matrix cores (3 products per term, fp32 accumulate);
// SSG_CONV_OUT_SPLIT (2) = out (and res) are written / read in h8l8.  flags = 0 is the plain fp32 path.
extern "C" int ssg_conv2d_nhwc_x(const void* in, const void* w, const float* bias, const void* res, void* out, int B, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, int flags, float acc_scale, const float* ch_scale,
                                 int32_t* overflow, hipStream_t stream) {
  ConvParams p;
  p.cscale = ch_scale; p.overflow = overflow;
  p.in = (const float*)in; p.w = (const float*)w; p.bias = bias; p.res = (const float*)res; p.out = (float*)out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.relu = relu;
  p.OH = (H + 2 * pad - KH) / stride + 1; p.OW = (W + 2 * pad - KW) / stride + 1;
  const int64_t M = (int64_t)B * p.OH * p.OW;
  const bool split = (flags & 1) != 0;
  if (B <= 0 || M <= 0 || M > 0x7fffffff || (Cout % 64) || !((Cin % 32) == 0 || Cin == 4) || stride < 1) {
    ssg_set_error("ssg_conv2d_nhwc: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d s=%d p=%d flags=%d", B, H, W, Cin, Cout, KH, KW, stride, pad, flags);
    return SSG_ERR_INVALID;
  }
  p.M = (int)M;
  const int64_t in_bytes = (int64_t)B * H * W * Cin * 4;
  if (in_bytes > 0x7fffffffLL) { ssg_set_error("ssg_conv2d_nhwc: input tensor of %lld bytes exceeds the 2 GiB buffer-resource range of this kernel; use a smaller batch", (long long)in_bytes); return SSG_ERR_INVALID; }
  p.in_bytes = (unsigned)in_bytes;
  p.in2 = nullptr; p.H2 = p.W2 = p.Cin2 = 0; p.stride2 = 1; p.in2_bytes = 0; p.rowterm = nullptr; p.epi = 0; p.tilemin = nullptr; p.tmin_ld = 0;
  p.out_split = (flags & 2) ? 1 : 0; p.res_split = p.out_split; p.acc_scale = acc_scale;
  static int variant = -1;
  if (variant < 0) { const char* e = getenv("SSG_CONV_VARIANT"); variant = e ? atoi(e) : 0; }
  p.variant = variant;
  const bool cin4 = (Cin == 4);
  p.Kpad = cin4 ? 32 * ((KH * KW + 7) / 8) : KH * KW * Cin;
  p.nk1 = p.Kpad / 16;   // no second input: every k-tile (of either BK) reads `in`
  if (cin4) return (Cout % 128 == 0) ? launch_conv<128, 128, 64, 64, true>(p, stream, split) : launch_conv<128, 64, 64, 32, true>(p, stream, split);
  if (conv_prefers_wide(p, split)) return launch_conv_wide(p, stream);
  return (Cout % 128 == 0 && !conv_prefers_bn64(p, split)) ? launch_conv<128, 128, 64, 64, false>(p, stream, split) : launch_conv<128, 64, 64, 32, false>(p, stream, split);
}

extern "C" int ssg_conv2d_nhwc_f32(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, hipStream_t stream) {
  return ssg_conv2d_nhwc_x(in, w, bias, res, out, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, 0, 1.f, nullptr, nullptr, stream);
}

// Fused bottleneck tail with a downsample branch (reid/models/base.py:75-90 when
// self.downsample is not None):  out = relu( conv3(in) + bn3 + downsample_conv(in2) + bn_ds )
// as ONE implicit GEMM over the concatenated K = Cin + Cin2: w [Cout][Cin + Cin2], bias = b3 + b_ds.
// in [B,H,W,Cin] (1x1, stride 1); in2 [B,H2,W2,Cin2] sampled at (oh*stride2, ow*stride2).
extern "C" int ssg_conv1x1_dual_nhwc_x(const void* in, const void* in2, const void* w, const float* bias, void* out, int B, int H, int W,
                                       int Cin, int H2, int W2, int Cin2, int stride2, int Cout, int relu, int flags, float acc_scale, const float* ch_scale,
                                       int32_t* overflow, hipStream_t stream) {
  ConvParams p;
  p.cscale = ch_scale; p.overflow = overflow;
  p.in = (const float*)in; p.w = (const float*)w; p.bias = bias; p.res = nullptr; p.out = (float*)out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.relu = relu;
  p.OH = H; p.OW = W;
  const int64_t M = (int64_t)B * H * W;
  const int64_t in_bytes = M * Cin * 4, in2_bytes = (int64_t)B * H2 * W2 * Cin2 * 4;
  if (B <= 0 || M > 0x7fffffff || (Cout % 64) || (Cin % 32) || (Cin2 % 32) || stride2 < 1 || (H - 1) * stride2 >= H2 || (W - 1) * stride2 >= W2 ||
      in_bytes > 0x7fffffffLL || in2_bytes > 0x7fffffffLL) {
    ssg_set_error("ssg_conv1x1_dual_nhwc: unsupported shape B=%d H=%d W=%d Cin=%d H2=%d W2=%d Cin2=%d s2=%d Cout=%d", B, H, W, Cin, H2, W2, Cin2, stride2, Cout);
    return SSG_ERR_INVALID;
  }
  p.M = (int)M; p.in_bytes = (unsigned)in_bytes;
  p.in2 = (const float*)in2; p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.stride2 = stride2; p.in2_bytes = (unsigned)in2_bytes;
  p.Kpad = Cin + Cin2; p.nk1 = Cin / 32; p.variant = 0; p.rowterm = nullptr; p.epi = 0; p.tilemin = nullptr; p.tmin_ld = 0;   // dual input stays on BK=32 (nk1 counts 32-wide tiles)
  p.out_split = (flags & 2) ? 1 : 0; p.res_split = p.out_split; p.acc_scale = acc_scale;
  const bool split = (flags & 1) != 0;
  if (conv_prefers_wide(p, split)) return launch_conv_wide(p, stream);
  return (Cout % 128 == 0 && !conv_prefers_bn64(p, split)) ? launch_conv<128, 128, 64, 64, false>(p, stream, split) : launch_conv<128, 64, 64, 32, false>(p, stream, split);
}

extern "C" int ssg_conv1x1_dual_nhwc_f32(const float* in, const float* in2, const float* w, const float* bias, float* out, int B, int H, int W,
                                         int Cin, int H2, int W2, int Cin2, int stride2, int Cout, int relu, hipStream_t stream) {
  return ssg_conv1x1_dual_nhwc_x(in, in2, w, bias, out, B, H, W, Cin, H2, W2, Cin2, stride2, Cout, relu, 0, 1.f, nullptr, nullptr, stream);
}

namespace ssg {
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int rows, int d, float scale, float* __restrict__ out) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= rows) return;
  const int lane = lane_id();
  float s = 0.f;
  const float* xr = x + (int64_t)row * d;
  // (the summation order -- lane c % 64 takes columns c, c + 64, ... in order, then the xor tree -- is part of the contract: the bound pass's
  // tolerance and the float32 distance epilogues were derived for it; only the loads are batched, eight per lane in flight)
  for (int c0 = lane; c0 < d; c0 += 512) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = xr[min(c0 + 64 * u, d - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) if (c0 + 64 * u < d) s += v[u] * v[u];
  }
  for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh, 64);
  if (lane == 0) out[row] = s * scale;
}
}  // namespace ssg

// Squared-L2 distance block in float32 (reid/evaluators.py:63-85 pairwise_distance):
//   out[i, j] = rowterm[i] + colterm[j] - 2 * <x_i, y_j>      x [m,d], y [n,d], out [m,n]
// rowterm/colterm are computed here: |x_i|^2 and |y_j|^2, or (self_form != 0, the reference's
// query=None branch :64-72) rowterm = 2*|x_i|^2, colterm = 0.  d % 32 == 0, n % 64 == 0 (pad).
// ws: m + n floats.  The fp32-MFMA instantiation of the convolution GEMM (SPLIT = false).
extern "C" int ssg_pairwise_sqdist_f32(const float* x, const float* y, int m, int n, int d, int self_form, float* ws, float* out, hipStream_t stream) {
  if (m <= 0 || n <= 0 || (d % 32) || (n % 64) || (int64_t)m * d * 4 > 0x7fffffffLL) {
    ssg_set_error("ssg_pairwise_sqdist_f32: need d %% 32 == 0, n %% 64 == 0 and x < 2 GiB (m=%d n=%d d=%d)", m, n, d);
    return SSG_ERR_INVALID;
  }
  float* rowterm = ws; float* colterm = ws + m;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((m + 3) / 4), dim3(256), 0, stream, x, m, d, self_form ? 2.f : 1.f, rowterm);
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, y, n, d, self_form ? 0.f : 1.f, colterm);
  ConvParams p;
  p.in = x; p.w = y; p.bias = colterm; p.res = nullptr; p.out = out;
  p.B = m; p.H = 1; p.W = 1; p.Cin = d; p.Cout = n; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.relu = 0; p.OH = 1; p.OW = 1;
  p.M = m; p.Kpad = d; p.nk1 = d / 16; p.variant = 0; p.in_bytes = (unsigned)((int64_t)m * d * 4);
  p.in2 = nullptr; p.H2 = p.W2 = p.Cin2 = 0; p.stride2 = 1; p.in2_bytes = 0; p.rowterm = rowterm; p.epi = 1; p.tilemin = nullptr; p.tmin_ld = 0; p.out_split = p.res_split = 0; p.acc_scale = 1.f;
  return (n % 128 == 0) ? launch_conv<128, 128, 64, 64, false>(p, stream) : launch_conv<128, 64, 64, 32, false>(p, stream);
}

// out[i,j] = 2 - 2 <x_i, y_j> in float32 (reid/rerank.py:174-182 original_dist of re_ranking_init).
// x [m,d], y [n,d], out [m,n]; d % 32 == 0, n % 64 == 0 (pad).  zeros [n] floats (bias slot).
extern "C" int ssg_cosine_dist_f32(const float* x, const float* y, int m, int n, int d, const float* zeros, float* out, hipStream_t stream) {
  if (m <= 0 || n <= 0 || (d % 32) || (n % 64) || (int64_t)m * d * 4 > 0x7fffffffLL) {
    ssg_set_error("ssg_cosine_dist_f32: need d %% 32 == 0, n %% 64 == 0 and x < 2 GiB (m=%d n=%d d=%d)", m, n, d);
    return SSG_ERR_INVALID;
  }
  ConvParams p;
  p.in = x; p.w = y; p.bias = zeros; p.res = nullptr; p.out = out;
  p.B = m; p.H = 1; p.W = 1; p.Cin = d; p.Cout = n; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.relu = 0; p.OH = 1; p.OW = 1;
  p.M = m; p.Kpad = d; p.nk1 = d / 16; p.variant = 0; p.in_bytes = (unsigned)((int64_t)m * d * 4);
  p.in2 = nullptr; p.H2 = p.W2 = p.Cin2 = 0; p.stride2 = 1; p.in2_bytes = 0; p.rowterm = nullptr; p.epi = 2; p.tilemin = nullptr; p.tmin_ld = 0; p.out_split = p.res_split = 0; p.acc_scale = 1.f;
  return (n % 128 == 0) ? launch_conv<128, 128, 64, 64, false>(p, stream) : launch_conv<128, 64, 64, 32, false>(p, stream);
}


extern "C" int ssg_nchw_to_nhwc4(const float* in, float* out, int B, int H, int W, int flip, hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0) { ssg_set_error("ssg_nchw_to_nhwc4: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(4096), dim3(256), 0, stream, in, out, B, H, W, flip);
  SSG_LAUNCH_CHECK("nchw_to_nhwc4_kernel");
  return SSG_OK;
}

extern "C" int ssg_nchw_to_nhwc4_h4l4(const float* in, void* out, int B, int H, int W, int flip, hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0) { ssg_set_error("ssg_nchw_to_nhwc4_h4l4: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(nchw_to_nhwc4_h4l4_kernel, dim3(4096), dim3(256), 0, stream, in, (float*)out, B, H, W, flip);
  SSG_LAUNCH_CHECK("nchw_to_nhwc4_h4l4_kernel");
  return SSG_OK;
}

extern "C" int ssg_maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t stream) {
  if (B <= 0 || (C & 3)) { ssg_set_error("ssg_maxpool3x3s2_nhwc: C %% 4 != 0"); return SSG_ERR_INVALID; }
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(8192), dim3(256), 0, stream, in, out, B, H, W, C, OH, OW);
  SSG_LAUNCH_CHECK("maxpool3x3s2_kernel");
  return SSG_OK;
}

extern "C" int ssg_maxpool3x3s2_h8l8(const void* in, void* out, int B, int H, int W, int C, hipStream_t stream) {
  if (B <= 0 || (C & 7)) { ssg_set_error("ssg_maxpool3x3s2_h8l8: C %% 8 != 0"); return SSG_ERR_INVALID; }
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_h8l8_kernel, dim3(8192), dim3(256), 0, stream, (const float*)in, (float*)out, B, H, W, C, OH, OW);
  SSG_LAUNCH_CHECK("maxpool3x3s2_h8l8_kernel");
  return SSG_OK;
}

extern "C" int ssg_gap_stripes_h8l8(const void* in, float* out, int B, int H, int W, int C, int num_split, hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || num_split > H) { ssg_set_error("ssg_gap_stripes_h8l8: bad shape"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(gap_stripes_h8l8_kernel, dim3(2048), dim3(256), 0, stream, (const float*)in, out, B, H, W, C, num_split);
  SSG_LAUNCH_CHECK("gap_stripes_h8l8_kernel");
  return SSG_OK;
}

extern "C" int ssg_h8l8_encode(const float* in, void* out, int64_t n, float scale, hipStream_t stream) {
  if (n <= 0 || (n & 7)) { ssg_set_error("ssg_h8l8_encode: n must be a positive multiple of 8"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(h8l8_encode_kernel, dim3(4096), dim3(256), 0, stream, in, (float*)out, n / 8, scale);
  SSG_LAUNCH_CHECK("h8l8_encode_kernel");
  return SSG_OK;
}

extern "C" int ssg_h8l8_decode(const void* in, float* out, int64_t n, float scale, hipStream_t stream) {
  if (n <= 0 || (n & 7)) { ssg_set_error("ssg_h8l8_decode: n must be a positive multiple of 8"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(h8l8_decode_kernel, dim3(4096), dim3(256), 0, stream, (const float*)in, out, n / 8, scale);
  SSG_LAUNCH_CHECK("h8l8_decode_kernel");
  return SSG_OK;
}

extern "C" int ssg_gap_stripes(const float* in, float* out, int B, int H, int W, int C, int num_split, hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || num_split > H) { ssg_set_error("ssg_gap_stripes: bad shape"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(gap_stripes_kernel, dim3(2048), dim3(256), 0, stream, in, out, B, H, W, C, num_split);
  SSG_LAUNCH_CHECK("gap_stripes_kernel");
  return SSG_OK;
}

extern "C" int ssg_flip_sum_l2norm(const float* a, const float* b, float* out, int rows, int C, hipStream_t stream) {
  if (rows <= 0 || C <= 0) { ssg_set_error("ssg_flip_sum_l2norm: empty"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(flip_sum_l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a, b, out, rows, C);
  SSG_LAUNCH_CHECK("flip_sum_l2norm_kernel");
  return SSG_OK;
}
