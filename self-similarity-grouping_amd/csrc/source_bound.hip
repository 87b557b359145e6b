// source_bound.hip -- bound pass of the source term (reid/rerank.py:36-37,39) as a plain fp16 GEMM.
//
// The source term needs, per target row, the MINIMUM over all sources of half(cdist(t, s)^2).  It is computed by
// filter-and-refine (conv.hip, ssg_source_rowmin_filtered): a matrix-core pass writes, per row and 8-source granule, a
// value that is within `tol` of the true squared distances of the granule; a float64 pass then re-evaluates only the
// granules that can still hold the row minimum.  The bound does not have to be accurate, only its error bounded -- so
// this pass multiplies the half-rounded operands directly (one v_mfma_f32_32x32x16_f16 per 32x32x16 block instead of the
// three of the split-half form) and, more importantly, reads 2 bytes per operand element instead of 4: the split-half
// kernel at these shapes is bound by L2 -> LDS traffic, not by the matrix pipe.
//   |x.y - half(x).half(y)| <= (2^-10 + 2^-22) sum_k |x_k y_k|  (+ fp32 accumulation), covered by the caller's tol.
//
// Kernel: D[n, m] = sum_k Y[n,k] X[m,k] on 128 x 128 tiles (4 waves x 64 x 64 = 2 x 2 MFMA tiles), BK = 32, register-staged
// double-buffered LDS (row pitch 80 B: conflict-free 16-byte fragment reads), epilogue = per target row the minimum of
// |x|^2 + |y|^2 - 2 x.y over every 8-source granule (same layout as conv.hip's epi 3, consumed by source_refine_kernel).
#include "ssg_common.h"

namespace ssg {
namespace sbound {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef SSG_SB_BK
#define SSG_SB_BK 32
#endif
constexpr int BM = 128, BN = 128, BK = SSG_SB_BK, PITCH = BK * 2 + 16;   // bytes per LDS row: BK halves + 16 of padding (pitch / 16 odd)
constexpr int CPR = BK * 2 / 16, NU = BM * CPR / 256;         // 16-byte chunks per row, chunks per thread and operand
constexpr int TILE_BYTES = BM * PITCH;                          // one operand tile of one stage

__global__ __launch_bounds__(256) void f32_to_f16_scaled_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int64_t n4, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    const v4h h = {(_Float16)(v.x * scale), (_Float16)(v.y * scale), (_Float16)(v.z * scale), (_Float16)(v.w * scale)};
    reinterpret_cast<v4h*>(out)[i] = h;
  }
}

// X [M, K] (targets), Y [Npad, K] (sources) half, K % BK == 0, Npad % 128 == 0.  tilemin [M, Npad/8].
__global__ __launch_bounds__(256, BK == 32 ? 4 : 2) void source_bound_kernel(const _Float16* __restrict__ X, const _Float16* __restrict__ Y, int M, int Npad, int K,
                                                              const float* __restrict__ rowterm, const float* __restrict__ colterm, float acc_scale,
                                                              float* __restrict__ tilemin, int tmin_ld) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TILE_BYTES];
  // Tile order for L2 reuse.  Workgroups are dealt to the 8 XCDs round-robin, each XCD has its own 4 MB L2: remap so that an
  // XCD owns a contiguous run of logical tiles, and order those in groups of TG target panels that sweep the source panels
  // together -- the TG target panels (TG x 512 KB at d = 2048) stay in that L2 and every source panel is fetched once per group
  // instead of once per target panel (the 53 MB of sources do not fit any L2).
  const int tiles_n = Npad / BN, tiles_m = (M + BM - 1) / BM;
  constexpr int TG = 4;
  int tm, tn;
  {
    const int nwg = tiles_m * tiles_n, b = (int)blockIdx.x;
    const int q = nwg / 8, r = nwg % 8, x = b % 8, sidx = b / 8;
    const int L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sidx;
    const int tpg = TG * tiles_n, grp = L / tpg, rem = L - grp * tpg;
    const int gcur = min(TG, tiles_m - grp * TG);
    tn = rem / gcur; tm = grp * TG + rem % gcur;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
  // staging: each operand tile of a stage is 128 rows x BK halves = 128 * CPR chunks of 16 bytes: NU per thread and operand
  const uint4* gx[NU]; const uint4* gy[NU]; int lo[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int c = tid + 256 * u, row = c / CPR, ch = c % CPR;
    gx[u] = reinterpret_cast<const uint4*>(X + (int64_t)min(tm * BM + row, M - 1) * K) + ch;
    gy[u] = reinterpret_cast<const uint4*>(Y + (int64_t)(tn * BN + row) * K) + ch;
    lo[u] = row * PITCH + ch * 16;
  }
  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  uint4 px[NU], py[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) { px[u] = gx[u][0]; py[u] = gy[u][0]; }
#pragma unroll
  for (int u = 0; u < NU; u++) { *reinterpret_cast<uint4*>(lds + lo[u]) = px[u]; *reinterpret_cast<uint4*>(lds + TILE_BYTES + lo[u]) = py[u]; }
  __syncthreads();
  for (int ks = 0; ks < nk; ks++) {
    const int nx = min(ks + 1, nk - 1) * CPR;                 // next stage's chunk offset (clamped: the last prefetch is repeated, unused)
#pragma unroll
    for (int u = 0; u < NU; u++) { px[u] = gx[u][nx]; py[u] = gy[u][nx]; }
    const unsigned char* xs = lds + (ks & 1) * (2 * TILE_BYTES);
    const unsigned char* ys = xs + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < BK / 16; s++) {       // 16-wide k steps of a stage
      v8h xf[2], yf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) xf[i] = *reinterpret_cast<const v8h*>(xs + (wm * 64 + i * 32 + l32) * PITCH + s * 32 + h * 16);
#pragma unroll
      for (int j = 0; j < 2; j++) yf[j] = *reinterpret_cast<const v8h*>(ys + (wn * 64 + j * 32 + l32) * PITCH + s * 32 + h * 16);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(yf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nk) {
      unsigned char* nb = lds + ((ks + 1) & 1) * (2 * TILE_BYTES);
#pragma unroll
      for (int u = 0; u < NU; u++) { *reinterpret_cast<uint4*>(nb + lo[u]) = px[u]; *reinterpret_cast<uint4*>(nb + TILE_BYTES + lo[u]) = py[u]; }
    }
    __syncthreads();
  }

  // epilogue.  D = Y * X^T: C/D layout col = lane&31 -> target row m, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> source n;
  // accumulator quad q of tile j holds sources j*32 + 8q + 4h + {0..3} of one target: the two half-waves together cover
  // the 8 sources of granule (j, q).
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = tm * BM + wm * 64 + i * 32 + l32;
    const float rt = rowterm[m < M ? m : 0];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n0 = tn * BN + wn * 64 + j * 32;
      float g[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 ct = *reinterpret_cast<const float4*>(colterm + n0 + 8 * q + 4 * h);
        const float mn = fminf(fminf((rt + ct.x) - 2.f * (acc[i][j][4 * q] * acc_scale), (rt + ct.y) - 2.f * (acc[i][j][4 * q + 1] * acc_scale)),
                               fminf((rt + ct.z) - 2.f * (acc[i][j][4 * q + 2] * acc_scale), (rt + ct.w) - 2.f * (acc[i][j][4 * q + 3] * acc_scale)));
        g[q] = fminf(mn, __shfl_xor(mn, 32, 64));
      }
      if (h == 0 && m < M) *reinterpret_cast<float4*>(tilemin + (int64_t)m * tmin_ld + n0 / 8) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

// ---- 256 x 256 tiles, LDS-DMA stages, 128 x 64 wave tiles ---------------------------------------------------------------------
// A one-product fp16 GEMM on 64 x 64 wave tiles reads one 16-byte fragment per lane and MFMA: at the matrix peak that alone is the
// CU's whole LDS bandwidth (1 KB per 32 cycles and wave, 4 SIMDs), before the staging writes -- the kernel above measured "fragment
// reads ~ MFMA time".  Here a wave owns 128 targets x 64 sources (6 fragments per 8 MFMAs), the operand tiles go global -> LDS
// directly (buffer_load ... lds: no staging registers, no ds_write; rows of 64 bytes = BK 32 halves, 16-byte chunks XOR-swizzled with
// (row >> 2) & 3 on the global side and in the fragment reads, as conv.hip's conv_dma_kernel does), four 32 KB stages with three
// k-tiles in flight, and the fragments of k-tile t + 1 are fetched into a second register set right after the barrier that publishes
// them, under the second half of tile t's MFMAs.  8 waves (2 per SIMD), 128 accumulator registers.
#define SSG_SB_LDSP(ptr_) ((__attribute__((address_space(3))) void*)(ptr_))
constexpr int TB = 256;
__global__ __launch_bounds__(512, 1) void source_bound_dma_kernel(const _Float16* __restrict__ X, const _Float16* __restrict__ Y, int M, int Npad, int K,
                                                                  const float* __restrict__ rowterm, const float* __restrict__ colterm, float acc_scale,
                                                                  float* __restrict__ tilemin, int tmin_ld, int gran4) {
  constexpr int NS = 4, XB = TB * 64, STAGE = 2 * XB, TDMA = 4, MT = 4, NT = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st2[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char st3[STAGE];
  const int tiles_n = (Npad + TB - 1) / TB, tiles_m = (M + TB - 1) / TB;
  constexpr int TG = 2;                                    // target panels (1 MB each at d = 2048) that sweep the source panels together
  int tm, tn;
  {
    const int nwg = tiles_m * tiles_n, b = (int)blockIdx.x;
    const int q = nwg / 8, r = nwg % 8, x = b % 8, sidx = b / 8;
    const int L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sidx;
    const int tpg = TG * tiles_n, grp = L / tpg, rem = L - grp * tpg;
    const int gcur = min(TG, tiles_m - grp * TG);
    tn = rem / gcur; tm = grp * TG + rem % gcur;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;
  // DMA: this wave fills rows [32 wave, 32 wave + 32) of the X part and of the Y part of every stage (two 16-row blocks each)
  const int drow = lane >> 2, pc = lane & 3, lc = pc ^ ((drow >> 2) & 3);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(X), 0, (unsigned)((int64_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Y), 0, (unsigned)((int64_t)Npad * K * 2), 0x00020000);
  // (rows past the end of X / Y lie outside the buffer range and read as zeros)
  const unsigned xo0 = (unsigned)(((int64_t)(tm * TB + wave * 32 + drow) * K) * 2 + lc * 16), xo1 = xo0 + (unsigned)(16 * K * 2);
  const unsigned yo0 = (unsigned)(((int64_t)(tn * TB + wave * 32 + drow) * K) * 2 + lc * 16), yo1 = yo0 + (unsigned)(16 * K * 2);
  const int nk = K / 32;
  int dn = 0;
#define SSG_SB_DMA(ST)                                                                                               \
  {                                                                                                                  \
    const unsigned kb_ = (unsigned)(dn * 64);                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, SSG_SB_LDSP(ST + wave * 2048), 16, xo0 + kb_, 0, 0, 0);              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, SSG_SB_LDSP(ST + wave * 2048 + 1024), 16, xo1 + kb_, 0, 0, 0);       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, SSG_SB_LDSP(ST + XB + wave * 2048), 16, yo0 + kb_, 0, 0, 0);         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, SSG_SB_LDSP(ST + XB + wave * 2048 + 1024), 16, yo1 + kb_, 0, 0, 0);  \
    dn += dn < nk - 1 ? 1 : 0;               /* the tail re-fetches the last tile (keeps the vmcnt accounting uniform) */ \
  }
  // fragments: lane (row l32 of a 32-row MFMA tile, k group h); k step s of a stage = logical 16-byte chunk 2 s + h
  const int g = (l32 >> 2) & 3;
  const int off0 = ((0 + h) ^ g) * 16, off1 = ((2 + h) ^ g) * 16;
  const int xrow = (wm * 128 + l32) * 64, yrow = XB + (wn * 64 + l32) * 64;
  v8h xf[2][2][MT], yf[2][2][NT];                           // [register set][k step][tile]
#define SSG_SB_READS(ST, S)                                                                                          \
  { _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                                 \
      xf[S][0][i] = *reinterpret_cast<const v8h*>(ST + xrow + i * 2048 + off0); xf[S][1][i] = *reinterpret_cast<const v8h*>(ST + xrow + i * 2048 + off1); } \
    _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                                 \
      yf[S][0][j] = *reinterpret_cast<const v8h*>(ST + yrow + j * 2048 + off0); yf[S][1][j] = *reinterpret_cast<const v8h*>(ST + yrow + j * 2048 + off1); } }
#define SSG_SB_MMA(S, KS) { _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < NT; j++) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(yf[S][KS][j], xf[S][KS][i], acc[i][j], 0, 0, 0); }
  // k-tile t (fragments in set S): first k step, publish tile t + 1 (and: everybody is done reading this stage), refill this stage with
  // tile t + NS, fetch tile t + 1's fragments into the other set, second k step
#define SSG_SB_STEP(ST, STN, S)                                                                                      \
  { SSG_SB_MMA(S, 0)                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TDMA * (NS - 2)) : "memory");                   \
    SSG_SB_DMA(ST)                                                                                                   \
    SSG_SB_READS(STN, 1 - (S))                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    SSG_SB_MMA(S, 1) }
  v16f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  SSG_SB_DMA(st0)
  SSG_SB_DMA(st1)
  SSG_SB_DMA(st2)
  SSG_SB_DMA(st3)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TDMA * (NS - 1)) : "memory");     // k-tile 0 landed for everybody
  SSG_SB_READS(st0, 0)
  const int nfull = nk / NS * NS;
  for (int kt = 0; kt < nfull; kt += NS) {
    SSG_SB_STEP(st0, st1, 0)
    SSG_SB_STEP(st1, st2, 1)
    SSG_SB_STEP(st2, st3, 0)
    SSG_SB_STEP(st3, st0, 1)
  }
  // left-over k-tiles (nk % NS): set 0 holds the fragments of tile nfull (published by the last barrier of the loop, or by the one above)
  if (nk - nfull >= 1) { SSG_SB_MMA(0, 0) SSG_SB_MMA(0, 1) }
  if (nk - nfull >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_SB_READS(st1, 0) SSG_SB_MMA(0, 0) SSG_SB_MMA(0, 1) }
  if (nk - nfull == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); SSG_SB_READS(st2, 0) SSG_SB_MMA(0, 0) SSG_SB_MMA(0, 1) }
#undef SSG_SB_STEP
#undef SSG_SB_MMA
#undef SSG_SB_READS
#undef SSG_SB_DMA
  // epilogue: as above (col = lane & 31 -> target, accumulator quad q of tile j = sources j * 32 + 8 q + 4 h + {0..3})
#pragma unroll
  for (int i = 0; i < MT; i++) {
    const int m = tm * TB + wm * 128 + i * 32 + l32;
    const float rt = rowterm[m < M ? m : 0];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int n0 = tn * TB + wn * 64 + j * 32;
      if (n0 >= Npad) continue;                              // (Npad is a multiple of 128, the tile of 256: the last source tile may be half empty)
      float mq[4], oq[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 ct = *reinterpret_cast<const float4*>(colterm + n0 + 8 * q + 4 * h);
        mq[q] = fminf(fminf((rt + ct.x) - 2.f * (acc[i][j][4 * q] * acc_scale), (rt + ct.y) - 2.f * (acc[i][j][4 * q + 1] * acc_scale)),
                      fminf((rt + ct.z) - 2.f * (acc[i][j][4 * q + 2] * acc_scale), (rt + ct.w) - 2.f * (acc[i][j][4 * q + 3] * acc_scale)));
        oq[q] = __shfl_xor(mq[q], 32, 64);                 // the other half-wave's four sources of the same 8
      }
      if (gran4) {
        // 4-source granules (round 4: the float64 pass re-reads half the bytes per candidate): granule (n0 + 8 q + 4 h) / 4 = n0 / 4 + 2 q + h;
        // half-wave 0 stores granules n0/4 + 0..3, half-wave 1 stores n0/4 + 4..7 -- one 16-byte store per lane, tmin_ld = Npad / 4
        const float4 w = h == 0 ? make_float4(mq[0], oq[0], mq[1], oq[1]) : make_float4(oq[2], mq[2], oq[3], mq[3]);
        if (m < M) *reinterpret_cast<float4*>(tilemin + (int64_t)m * tmin_ld + n0 / 4 + 4 * h) = w;
      } else if (h == 0 && m < M) {
        *reinterpret_cast<float4*>(tilemin + (int64_t)m * tmin_ld + n0 / 8) = make_float4(fminf(mq[0], oq[0]), fminf(mq[1], oq[1]), fminf(mq[2], oq[2]), fminf(mq[3], oq[3]));
      }
    }
  }
}

}  // namespace sbound
}  // namespace ssg
