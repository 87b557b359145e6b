// source_term.hip -- K4, the source-domain term of reid/rerank.py:35-40 by filter-and-refine: host side of the bound pass (the GEMM kernels are
// conv.hip's split-half kernel or source_bound.hip's one-product LDS-DMA kernel) and the float64 refine kernel.  Moved out of conv.hip in
// round 4 (conv.hip holds the embedding's convolution kernels: its text is part of the fingerprint that ties a PMC traffic summary to a build).
#include "ssg_common.h"

namespace ssg {
// Exact refinement of the source-term minimum: one wave per target row.  Granules (8 sources) whose float32
// lower-bounded minimum can still beat the row's best are re-evaluated in float64 with the difference form
// sum_k (x_k - y_k)^2 (like cdist), then half(sqrt(s)^2) as in reid/rerank.py:36-37.
// G = sources per granule (8: every bound kernel; 4: source_bound_dma_kernel with gran4), 64 / G lanes share one source's k range.
template <int G>
__global__ __launch_bounds__(256) void source_refine_kernel(const float* __restrict__ tgt, const float* __restrict__ src, const float* __restrict__ tilemin,
                                                            int ld, int ngran, float tol, int nrows, int Ns, int d, unsigned* __restrict__ rowmin) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= nrows) return;
  const int lane = lane_id();
  float gmin = INFINITY;
  const float* tm = tilemin + (int64_t)row * ld;
  for (int t0 = lane; t0 < ngran; t0 += 512) {          // eight bounds per lane in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = tm[min(t0 + 64 * u, ngran - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) gmin = fminf(gmin, v[u]);
  }
  for (int sh = 1; sh < 64; sh <<= 1) gmin = fminf(gmin, __shfl_xor(gmin, sh, 64));
  const float bound = gmin + tol;
  const float* x = tgt + (int64_t)row * d;
  unsigned best = 0xffffffffu;
  for (int tb = 0; tb < ngran; tb += 512) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = tm[min(tb + 64 * u + lane, ngran - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) {
    const int t0 = tb + 64 * u;
    if (t0 >= ngran) break;
    const int t = t0 + lane;
    uint64_t cand = __ballot(t < ngran && v[u] <= bound);
    while (cand) {
      const int tt = t0 + __ffsll((long long)cand) - 1;
      cand &= cand - 1;
      // G sources of the granule: lane group g = lane / LPS takes source tt*G+g, its LPS = 64 / G lanes split k
      constexpr int LPS = 64 / G;
      const int sidx = tt * G + lane / LPS;
      double acc = 0.0;
      if (sidx < Ns) {
        const float* y = src + (int64_t)sidx * d;
        // eight 16-byte pieces of x and of y in flight per lane (round 4: one pair per trip made every candidate a chain of 64 L2 / fabric
        // round trips -- the pass ran at the latency, not at the bandwidth, of the 128 KB it reads per row); same summation order
        for (int k0 = (lane & (LPS - 1)) * 4; k0 < d; k0 += 32 * LPS) {
          float4 xv[8], yv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int k = min(k0 + 4 * LPS * u, d - 4);
            xv[u] = *reinterpret_cast<const float4*>(x + k); yv[u] = *reinterpret_cast<const float4*>(y + k);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (k0 + 4 * LPS * u >= d) break;
            const double d0 = (double)xv[u].x - (double)yv[u].x, d1 = (double)xv[u].y - (double)yv[u].y, d2 = (double)xv[u].z - (double)yv[u].z,
                         d3 = (double)xv[u].w - (double)yv[u].w;
            acc += d0 * d0; acc += d1 * d1; acc += d2 * d2; acc += d3 * d3;
          }
        }
      }
      for (int sh = 1; sh < LPS; sh <<= 1) acc += __shfl_xor(acc, sh, 64);
      if (sidx < Ns) {
        const double dist = sqrt(acc);
        const unsigned hb = d2h(dist * dist);     // np.power(cdist, 2).astype(float16)
        best = best < hb ? best : hb;
      }
    }
    }
  }
  for (int sh = 1; sh < 64; sh <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)best, sh, 64); best = best < o ? best : o; }
  if (lane == 0) rowmin[row] = best;
}
}  // namespace ssg

// round 5: |x|^2 of every row AND the scaled half copy the one-product bound pass multiplies, in ONE pass over the features (they were two
// launches per operand set: 237 MB read twice).  The squared norm is summed in exactly row_sqnorm_kernel's order (lane c % 64 takes columns
// c, c + 64, ... in order, then the xor tree: the bound pass's tolerance and the refine pass were derived for it) and the halves are the
// values f32_to_f16_scaled_kernel writes, so every result downstream is unchanged.
__global__ __launch_bounds__(256) void encode_sqnorm_kernel(const float* __restrict__ x, int rows, int d, float scale, _Float16* __restrict__ x16,
                                                            float* __restrict__ out) {
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= rows) return;
  const int lane = lane_id();
  float s = 0.f;
  const float* xr = x + (int64_t)row * d;
  _Float16* hr = x16 + (int64_t)row * d;
  for (int c0 = lane; c0 < d; c0 += 512) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = xr[min(c0 + 64 * u, d - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) if (c0 + 64 * u < d) { s += v[u] * v[u]; hr[c0 + 64 * u] = (_Float16)(v[u] * scale); }
  }
  for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh, 64);
  if (lane == 0) out[row] = s;
}

// Source-term row minimum (reid/rerank.py:36-37,39) by filter-and-refine: a float32 MFMA pass bounds every
// target-source distance (per row and 8-source granule), a float64 pass re-evaluates only the granules within `tol`
// of the row's bound.  Same result as ssg_source_rowmin_f16 (exact min of the half-rounded float64 distances)
// whenever tol >= the float32 error of the bound (callers pass 8*d*2^-24*max|x|*max|y| + margin).
// tgt [nrows,d], src [Ns_pad,d] (rows >= Ns are padding), d % 32 == 0, Ns_pad % 128 == 0.
// ws: nrows + Ns_pad + nrows*(Ns_pad/8) floats, plus (nrows + Ns_pad)*d floats when scale_t, scale_s > 0: the bound
// pass then runs on the fp16 matrix cores over split-half copies of tgt*scale_t and src*scale_s (powers of two that
// keep max|x|*scale < 65504); the caller's tol must cover that pass's error (3 products, 3d-term accumulation).
static int source_rowmin_filtered_impl(const float* tgt, const float* src, int nrows, int Ns, int Ns_pad, int d, float tol, float scale_t, float scale_s,
                                       int one_product, float* ws, uint32_t* rowmin, hipStream_t stream) {
  if (nrows <= 0 || Ns <= 0 || Ns_pad < Ns || (Ns_pad % 128) || (d % 32) || (int64_t)nrows * d * 4 > 0x7fffffffLL) {
    ssg_set_error("ssg_source_rowmin_filtered: bad shape nrows=%d Ns=%d Ns_pad=%d d=%d", nrows, Ns, Ns_pad, d);
    return SSG_ERR_INVALID;
  }
  float* rowterm = ws; float* colterm = ws + nrows; float* tilemin = colterm + Ns_pad;
  int ntiles = Ns_pad / 8;   // 8-source granules
  const bool split = scale_t > 0.f && scale_s > 0.f;
  const bool one = split && one_product && (Ns_pad % 128) == 0 && (d % sbound::BK) == 0;
  if (!one) {        // (the one-product path computes the norms together with its half copies, below)
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, nrows, d, 1.f, rowterm);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((Ns_pad + 3) / 4), dim3(256), 0, stream, src, Ns_pad, d, 1.f, colterm);
    if (Ns_pad > Ns) SSG_HIP(hipMemsetAsync(colterm + Ns, 0x7f, (size_t)(Ns_pad - Ns) * sizeof(float), stream));   // 0x7f7f7f7f = 3.4e38: padding never wins
  }
  if (one) {
    // bound pass as a plain fp16 GEMM on half copies of the scaled operands (source_bound.hip): 2 bytes per element, 1 product
    static int sb_dma = -1;                 // SSG_SB_DMA=0: the register-staged 128 x 128 kernel
    if (sb_dma < 0) { const char* e = getenv("SSG_SB_DMA"); sb_dma = e ? atoi(e) : 1; }
    static int sb_gran = -1;                // SSG_SB_GRAN=8: 8-source granules everywhere
    if (sb_gran < 0) { const char* e = getenv("SSG_SB_GRAN"); sb_gran = e ? atoi(e) : 4; }
    // 4-source granules halve what the float64 pass re-reads per candidate (it runs at the L2 <- fabric rate: 0.46 -> 0.26 ms at the bench's
    // shape) and double the bounds table; used when that table still fits the documented workspace next to the half copies (which need
    // only half of the (nrows + Ns_pad) * d floats the split-half copies would take): always at d = 2048, not for short features
    const int64_t doc_floats = (int64_t)nrows + Ns_pad + (int64_t)nrows * (Ns_pad / 8) + ((int64_t)nrows + Ns_pad) * d;
    const int64_t need4 = (int64_t)nrows + Ns_pad + (int64_t)nrows * (Ns_pad / 4) + 4 + (((int64_t)nrows + Ns_pad) * d + 1) / 2;
    const int gran4 = (sb_dma && sb_gran == 4 && need4 <= doc_floats && (int64_t)nrows * d * 2 < 0x7fffffffLL && (int64_t)Ns_pad * d * 2 < 0x7fffffffLL) ? 1 : 0;
    if (gran4) ntiles = Ns_pad / 4;
    uintptr_t a = (uintptr_t)(tilemin + (int64_t)nrows * ntiles); a = (a + 15) & ~(uintptr_t)15;
    _Float16* x16 = (_Float16*)a; _Float16* y16 = x16 + (int64_t)nrows * d;
    static int fused_enc = -1;              // SSG_SB_FUSED_ENC=0: separate norm and conversion launches (rounds 2-4)
    if (fused_enc < 0) { const char* e = getenv("SSG_SB_FUSED_ENC"); fused_enc = e ? atoi(e) : 1; }
    if (fused_enc) {
      hipLaunchKernelGGL(encode_sqnorm_kernel, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, nrows, d, scale_t, x16, rowterm);
      hipLaunchKernelGGL(encode_sqnorm_kernel, dim3((Ns_pad + 3) / 4), dim3(256), 0, stream, src, Ns_pad, d, scale_s, y16, colterm);
    } else {
      hipLaunchKernelGGL(row_sqnorm_kernel, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, nrows, d, 1.f, rowterm);
      hipLaunchKernelGGL(row_sqnorm_kernel, dim3((Ns_pad + 3) / 4), dim3(256), 0, stream, src, Ns_pad, d, 1.f, colterm);
      hipLaunchKernelGGL(sbound::f32_to_f16_scaled_kernel, dim3(4096), dim3(256), 0, stream, tgt, x16, (int64_t)nrows * d / 4, scale_t);
      hipLaunchKernelGGL(sbound::f32_to_f16_scaled_kernel, dim3(4096), dim3(256), 0, stream, src, y16, (int64_t)Ns_pad * d / 4, scale_s);
    }
    if (Ns_pad > Ns) SSG_HIP(hipMemsetAsync(colterm + Ns, 0x7f, (size_t)(Ns_pad - Ns) * sizeof(float), stream));   // 0x7f7f7f7f = 3.4e38: padding never wins
    if (sb_dma && (int64_t)nrows * d * 2 < 0x7fffffffLL && (int64_t)Ns_pad * d * 2 < 0x7fffffffLL) {     // (operands go through 2 GiB buffer resources)
      const int tiles = ((nrows + sbound::TB - 1) / sbound::TB) * ((Ns_pad + sbound::TB - 1) / sbound::TB);
      hipLaunchKernelGGL(sbound::source_bound_dma_kernel, dim3(tiles), dim3(512), 0, stream, x16, y16, nrows, Ns_pad, d, rowterm, colterm,
                         1.f / (scale_t * scale_s), tilemin, ntiles, gran4);
      if (gran4) {
        hipLaunchKernelGGL(source_refine_kernel<4>, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, src, tilemin, ntiles, ntiles, tol, nrows, Ns, d, rowmin);
        SSG_LAUNCH_CHECK("source_bound / source_refine kernels");
        return SSG_OK;
      }
    } else {
      const int tiles = ((nrows + sbound::BM - 1) / sbound::BM) * (Ns_pad / sbound::BN);
      hipLaunchKernelGGL(sbound::source_bound_kernel, dim3(tiles), dim3(256), 0, stream, x16, y16, nrows, Ns_pad, d, rowterm, colterm,
                         1.f / (scale_t * scale_s), tilemin, ntiles);
    }
    hipLaunchKernelGGL(source_refine_kernel<8>, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, src, tilemin, ntiles, ntiles, tol, nrows, Ns, d, rowmin);
    SSG_LAUNCH_CHECK("source_bound / source_refine kernels");
    return SSG_OK;
  }
  const float* gt = tgt; const float* gs = src;
  if (split) {   // bound pass on the fp16 matrix cores: split-half copies of both operand sets (scaled into the half range)
    float* ts = tilemin + (int64_t)nrows * ntiles; float* ss = ts + (int64_t)nrows * d;
    hipLaunchKernelGGL(h8l8_encode_kernel, dim3(4096), dim3(256), 0, stream, tgt, ts, (int64_t)nrows * d / 8, scale_t);
    hipLaunchKernelGGL(h8l8_encode_kernel, dim3(4096), dim3(256), 0, stream, src, ss, (int64_t)Ns_pad * d / 8, scale_s);
    gt = ts; gs = ss;
  }
  ConvParams p;
  p.in = gt; p.w = gs; p.bias = colterm; p.res = nullptr; p.out = nullptr;
  p.B = nrows; p.H = 1; p.W = 1; p.Cin = d; p.Cout = Ns_pad; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.relu = 0; p.OH = 1; p.OW = 1;
  p.M = nrows; p.Kpad = d; p.nk1 = d / 16; p.variant = 0; p.in_bytes = (unsigned)((int64_t)nrows * d * 4);
  p.in2 = nullptr; p.H2 = p.W2 = p.Cin2 = 0; p.stride2 = 1; p.in2_bytes = 0; p.rowterm = rowterm; p.epi = 3; p.tilemin = tilemin; p.tmin_ld = ntiles; p.out_split = p.res_split = 0;
  p.acc_scale = split ? 1.f / (scale_t * scale_s) : 1.f;
  // 128x256 tiles when the padded source count allows: 3/4 of the global->LDS bytes of the 128x128 tile
  int rc = split ? ((Ns_pad % 256) == 0 ? launch_conv_wide(p, stream) : launch_conv_bk<128, 128, 64, 64, false, 32, true>(p, stream))
                 : launch_conv<128, 128, 64, 64, false>(p, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(source_refine_kernel<8>, dim3((nrows + 3) / 4), dim3(256), 0, stream, tgt, src, tilemin, ntiles, ntiles, tol, nrows, Ns, d, rowmin);
  SSG_LAUNCH_CHECK("source_refine_kernel");
  return SSG_OK;
}

extern "C" int ssg_source_rowmin_filtered(const float* tgt, const float* src, int nrows, int Ns, int Ns_pad, int d, float tol, float scale_t, float scale_s,
                                          float* ws, uint32_t* rowmin, hipStream_t stream) {
  return source_rowmin_filtered_impl(tgt, src, nrows, Ns, Ns_pad, d, tol, scale_t, scale_s, 0, ws, rowmin, stream);
}
// the same with the bound pass on the hi halves only (one fp16 product per term instead of three: 1/3 of the matrix work); the
// caller's tol must cover 2^-10 |x||y| per dot product on top of the accumulation error.  Needs scale_t, scale_s > 0 and
// Ns_pad % 256 == 0, otherwise it runs the three-product pass.
extern "C" int ssg_source_rowmin_filtered1(const float* tgt, const float* src, int nrows, int Ns, int Ns_pad, int d, float tol, float scale_t, float scale_s,
                                           float* ws, uint32_t* rowmin, hipStream_t stream) {
  return source_rowmin_filtered_impl(tgt, src, nrows, Ns, Ns_pad, d, tol, scale_t, scale_s, 1, ws, rowmin, stream);
}
