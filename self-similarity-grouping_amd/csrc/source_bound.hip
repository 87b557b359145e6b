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

}  // namespace sbound
}  // namespace ssg
