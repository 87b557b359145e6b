// bottleneck.hip -- one bottleneck block of the ResNet-50 backbone in ONE kernel (split-half path): the three blocks of layer1
// and the three identity blocks of layer2.
//
// reid/models/base.py:57-90 (torchvision Bottleneck without a downsample branch), eval mode, BatchNorm folded:
//   out = relu( conv3_1x1( relu( conv2_3x3( relu( conv1_1x1(x) ) ) ) ) + x )
// Run as three launches (conv.hip) a layer1 block moves 16 "units" of 64-channel activations through HBM (x is read by
// conv1 and again as the residual, the two 64-channel intermediates are written and re-read) and its three GEMMs have short
// reductions (K = 256 / 576 / 64): the launches sit at 4.4-4.9 TB/s or stall on their pipeline fill.  Here a workgroup owns
// TH full-width image rows: it computes conv1 on TH+2 rows (one halo row above and below; the left/right halo is the zero
// padding of conv2, never computed), keeps that intermediate in LDS in the split-half format, runs conv2 as an implicit GEMM
// whose pixel operand is read from LDS (the 9 taps are 9 shifted fragment addresses), turns conv2's accumulators through LDS
// again into the pixel operand of conv3 and finishes with bias + residual + ReLU + re-encoding.  Measured (TCC_EA0_RDREQ): the
// L2 fetches 2.07 x |x| per launch -- x once, the halo rows mostly L2 hits of the neighbouring workgroup, and the residual
// re-read (14 us after phase 1, 64 workgroups streaming through each 4 MB L2) once more from HBM -- and writes |out| once:
// 3.1 units against the 16 of the three launches.  (Keeping the residual lines recent with touch loads during phase 2 does not
// work: the touches already miss.)
//
// Numerics: the same three-product split-half multiply, the same reduction order and the same epilogues as the separate
// launches (conv.hip), so the block output is bit-identical to the three-launch path.
//
// DS variant (the first block of layer1, base.py:75-90 with a stride-1 downsample branch): x has CIN = 64 channels, the
// downsample 1x1 convolution rides in conv3's reduction (K = MID + CIN, weights concatenated, as ssg_conv1x1_dual_nhwc_x does)
// and replaces the residual; its pixel operand (this wave's 32 pixels x CIN channels) is loaded straight into MFMA fragments.
//
// Layer1 (C = 256, MID = 64, 32-pixel rows, TH = 4): 4 waves, 2 workgroups per CU (<= 80 KB of LDS each) -- one's loads run under
// the other's multiplies.  Layer2 (C = 512, MID = 128, 16-pixel rows, TH = 8): the 128-channel halo intermediate alone is 85 KB,
// so ONE 8-wave workgroup per CU (150 KB): phase 2 = 4 pixel tiles x 2 channel groups, phase 3 = a wave pair per pixel tile, the
// epilogue patches take over the dead W3 stages; its phases are serial on a CU (measured 0.75 -> 0.62 ms per block).
// All operand tiles are staged global -> registers -> LDS with the loads several k-tiles ahead (register rings), one barrier per
// stage:
//   phase 1  y1[(TH+2)*IW, MID] = relu(x W1^T)   k-tiles of 16 channels: x rows (64 B each) + W1 rows; wave = 3 x 1 MFMA tiles
//   phase 2  y2[TH*IW, MID]     = relu(conv3x3(y1))   k-tiles = (32-channel chunk, tap): W2 rows (128 B); wave = 32 pixels x MID
//   phase 3  out[TH*IW, C]      = relu(y2 W3^T + x)   k-tiles of 16 channels: W3 rows; wave = its own 32 pixels x all C channels
#include "ssg_common.h"
#include <cstdlib>

// Round 6: cache policy of the block's two streams that nobody re-reads from the L2 -- the stores of `out` (1-2 GB per launch at B = 1000)
// and the residual re-read of x (its last use).  With the default policy both allocate lines in the 4 MB L2s that the x rows, the halo
// rows of the neighbouring workgroup and the weights go through; with `nt` the layer1 identity blocks run 1.48 -> 1.30-1.35 ms and the
// layer2 ones 1.22 -> 1.12 ms at nearly the same fabric traffic (FETCH_SIZE - 3 %: profiles/r06_ab_nt_policy.txt).  Same values, same
// order: bit-identical.
#ifndef SSG_BN_NT_STORE
#define SSG_BN_NT_STORE 1
#endif
#ifndef SSG_BN_NT_RES
#define SSG_BN_NT_RES 1
#endif

namespace ssg {
namespace bneck {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));   // staging registers: arrays of HIP's float4 struct stay in scratch memory

struct Params {
  const float* x; float* out;
  const float* w1; const float* b1; const float* cs1;   // [MID][C]      h8l8, rows pre-scaled by 1/cs1
  const float* w2; const float* b2; const float* cs2;   // [MID][9*MID]  k = ((c/32)*9 + tap)*32 + c%32
  const float* w3; const float* b3; const float* cs3;   // [C][MID]
  int B, H;
  int* overflow;
  unsigned long long* prof = nullptr;   // SSG_BN_PROF builds: 8 phase timestamps (s_memtime) per workgroup
};
#ifdef SSG_BN_PROF
#define SSG_BN_STAMP(I_) { if (p.prof && threadIdx.x == 0) p.prof[(size_t)blockIdx.x * 8 + (I_)] = __builtin_readcyclecounter(); }
#else
#define SSG_BN_STAMP(I_)
#endif

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int C, int MID, int IW, int TH, int CIN = C, int NW = 4>
struct Cfg {
  static constexpr int HROWS = TH + 2, NPIX1 = HROWS * IW, NPIX = TH * IW;
  static constexpr int NTHR = NW * 64, RPP = NTHR / 4;  // 64-byte k-tile rows staged per pass of the workgroup (phase 3)
  // phase-1 k-tiles: 16 channels (64-byte row pieces) or, for the 8-wave layer2 kernel, 32 channels (whole 128-byte lines, half the
  // load -> LDS -> barrier -> fragment round trips: a step costs ~1250 cycles in those round trips alone, whatever it multiplies)
#ifndef SSG_BN_K1T
#define SSG_BN_K1T 32
#endif
#ifndef SSG_BN_K1T4
#define SSG_BN_K1T4 32               // round 4: whole 128-byte lines per pixel row and k-tile for the 4-wave layer1 kernel too (16: 64-byte pieces,
#endif                               // two texture-addresser line accesses per useful line): identity blocks 1.59 -> 1.53 ms, bit-identical
  static constexpr int K1T = NW == 8 ? SSG_BN_K1T : SSG_BN_K1T4, KS1 = K1T / 16, CPR1 = 4 * KS1, RPP1 = NTHR / CPR1;
  static constexpr int XROWS = (NPIX1 + RPP1 - 1) / RPP1 * RPP1;   // x rows of a phase-1 stage, padded to whole passes (the padding loads return zeros)
  static constexpr int P1 = 80;                        // LDS pitch of a 64-byte k-tile row (pitch/16 odd: conflict-free b128)
  static constexpr int P1T = K1T * 4 + 16;             // ... of a phase-1 row (80 or 144 bytes)
  static constexpr int KT2 = NW == 8 ? 2 : 1;          // 32-channel (chunk, tap) k-tiles of conv2 per LDS stage: the 8-wave workgroup (alone on its CU) halves its barriers
  static constexpr int P2 = KT2 * 128 + 16;            // ... of a phase-2 stage row (KT2 x 128 bytes of a W2 row)
  static constexpr int PY = MID * 4 + 16;              // ... of a y1 / y2 pixel row (all MID channels, h8l8)
  static constexpr int BUF1 = (XROWS + MID) * P1T;     // phase-1 stage: x rows, then W1 rows
  static constexpr int ZERO_OFF = NPIX1 * PY;          // one all-zero pixel row (conv2's left / right padding)
  static constexpr int W2_OFF = (ZERO_OFF + PY + 255) / 256 * 256, BUF2 = MID * P2;
  static constexpr int W3_OFF = (NPIX * PY + 255) / 256 * 256, BUF3 = C * P1;
  static constexpr int LDS = cmax(cmax(2 * BUF1, W2_OFF + 2 * BUF2), W3_OFF + 2 * BUF3);
  static constexpr int DS = CIN != C;                  // downsample variant: conv3's reduction is [y2 | x]
#ifdef SSG_BN_ABL_NK1                                  // ablation builds (tools/micro/bneck_prof.hip): a shortened conv1 reduction, wrong results
  static constexpr int NK1 = SSG_BN_ABL_NK1;
#else
  static constexpr int NK1 = CIN / K1T;
#endif
  static constexpr int NK2 = (MID / 32) * 9 / KT2, NK3 = (MID + (DS ? CIN : 0)) / 16;
  // stages of global loads in flight ahead of the multiply (PD1 = 4 measures the same: phase 1 is bound by HBM bandwidth, not latency)
  static constexpr int PD1MAX = K1T == 64 ? 2 : (K1T == 32 ? 4 : 8);
  static constexpr int PD1 = NK1 < PD1MAX ? NK1 : PD1MAX, PD2 = KT2 == 1 ? 6 : 3;
  static_assert(((MID / 32) * 9) % KT2 == 0 && NK2 % PD2 == 0, "phase-2 stages and ring");
  static_assert((NW == 4 || NW == 8) && NPIX == 128 && NPIX1 % 32 == 0 && MID % RPP == 0 && MID % RPP1 == 0 && C % RPP == 0 && MID % (NW * 8) == 0, "128 output pixels, whole staging passes");
  static constexpr bool ZERO_EARLY = 2 * BUF1 <= ZERO_OFF;   // the zero row is written while phase 1 runs -- unless the (32-channel) stages cover it
  static_assert(LDS <= (NW == 4 ? 80 : 160) * 1024, "4 waves: two workgroups per CU; 8 waves: one");
  static_assert(NW == 4 || NW * 32 * 36 * 4 <= 2 * BUF3, "8 waves: the epilogue patches take over the W3 stages");
};

__device__ __forceinline__ unsigned pack2(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ void encode4(const float4 v, uint2& hi, uint2& lo) {
  const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
  hi = make_uint2(pack2(h0, h1), pack2(h2, h3));
  lo = make_uint2(pack2((_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1)), pack2((_Float16)(v.z - (float)h2), (_Float16)(v.w - (float)h3)));
}
// bits 15 / 31 of the result are set when a hi half is infinite or NaN (exponent field all ones: + 0x0400 carries into the sign
// position, nothing smaller does); OR-accumulated over the kernel and tested once at the end
__device__ __forceinline__ unsigned hi_nonfinite_bits(const uint2 hi) {
  return ((hi.x & 0x7c007c00u) + 0x04000400u) | ((hi.y & 0x7c007c00u) + 0x04000400u);
}
__device__ __forceinline__ float hlo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float hhi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
__device__ __forceinline__ float4 decode4(const uint2 hi, const uint2 lo) {
  return make_float4(hlo(hi.x) + hlo(lo.x), hhi(hi.x) + hhi(lo.x), hlo(hi.y) + hlo(lo.y), hhi(hi.y) + hhi(lo.y));
}
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
}
// x*w = xh*wl + xl*wh + xh*wh (the order of conv.hip's split_mma_step), weights as the first MFMA operand:
// C/D layout col = lane&31 -> pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> channel

// one LDS-DMA instruction: 64 lanes x 16 bytes, global (buffer resource, per-lane byte offset + uniform offset) -> LDS at `lds_addr` +
// 16 * lane.  A function of its own: with the builtin called directly from the kernel template (per-lane offset from a local array) the
// HOST pass of hipcc drops the kernel's launch stub without a diagnostic (undefined __device_stub__ at load time).
__device__ __forceinline__ void bn_dma16(const __amdgpu_buffer_rsrc_t r, unsigned lds_addr, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, voff, soff, 0, 0);
}

template <int C, int MID, int IW, int TH, int CIN, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void bottleneck_kernel(Params p) {
  using K = Cfg<C, MID, IW, TH, CIN, NW>;
  constexpr bool DS = K::DS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
  const int tiles_img = p.H / TH, ntiles = p.B * tiles_img;
  int T;
  {   // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous run of tiles (halo rows hit its L2)
    const int b = (int)blockIdx.x, q = ntiles / 8, r = ntiles % 8, x = b % 8, s = b / 8;
    T = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
  }
  const int img = T / tiles_img, ty0 = (T - img * tiles_img) * TH;
  SSG_BN_STAMP(0)
  if (K::ZERO_EARLY && tid < K::PY / 16) *reinterpret_cast<uint4*>(smem + K::ZERO_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // =========================== phase 1: y1 = relu(conv1(x)) on the TH+2 halo rows ===========================
  constexpr int RPP = K::RPP, RPP1 = K::RPP1, AU = K::XROWS / RPP1, WU = MID / RPP1;   // 16-byte pieces per thread and k-tile: x rows, W1 rows
  const int ck = tid & 3, r0 = tid >> 2;               // phase 3 staging (64-byte rows)
  const int ck1 = tid % K::CPR1, r1 = tid / K::CPR1;   // phase 1 staging (64- or 128-byte rows)
  const float* ximg = p.x + (int64_t)img * p.H * IW * CIN;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, (unsigned)(p.H * IW * CIN * 4), 0x00020000);
  // MFMA tiles of phase 1: (NPIX1/32) pixel tiles x (MID/32) channel tiles over the waves
  constexpr int MT1 = K::NPIX1 / 32, NT1 = MID / 32;
  // waves are split as WR x WC with WC = min(NT1, NW/2): each wave owns ceil(MT1/WR) pixel tiles x NT1/WC channel tiles (a pixel tile
  // beyond MT1 multiplies the zero padding rows of the stage and is not written)
  constexpr int WC1 = NT1 < NW / 2 ? NT1 : NW / 2, WR1 = NW / WC1, MTW1 = (MT1 + WR1 - 1) / WR1, NTW1 = NT1 / WC1;
  static_assert(NT1 % WC1 == 0 && (WR1 * MTW1) * 32 <= K::XROWS, "phase-1 tiles divide over the waves");
  const int i1b = (wave / WC1) * MTW1, j1b = (wave % WC1) * NTW1;
  v16f acc1[MTW1][NTW1];
#pragma unroll
  for (int i = 0; i < MTW1; i++)
#pragma unroll
    for (int j = 0; j < NTW1; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[i][j][r] = 0.f;
  // folded BatchNorm scale / bias of this wave's conv1 channels (needed after the loop: no L2 round trip there)
  float4 cs1r[NTW1][4], b1r[NTW1][4];
#pragma unroll
  for (int j = 0; j < NTW1; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      cs1r[j][q] = *reinterpret_cast<const float4*>(p.cs1 + (j1b + j) * 32 + 8 * q + 4 * h); b1r[j][q] = *reinterpret_cast<const float4*>(p.b1 + (j1b + j) * 32 + 8 * q + 4 * h);
    }
#ifndef SSG_BN_P1DMA
#define SSG_BN_P1DMA 1
#endif
  if constexpr (NW == 8 && K::K1T == 32 && SSG_BN_P1DMA) {
    // ---- 8-wave kernel: LDS-DMA stages + register double-buffered fragments.  One workgroup per CU: with register-staged stages every
    // wave stores, waits at the barrier, reads its fragments and multiplies at the same time as the other seven -- per 32-channel k-tile
    // 1350 cycles of LDS traffic and 1150 cycles of MFMA one after the other (measured 3000).  Here the x rows and W1 rows go
    // global -> LDS directly (buffer_load ... lds; rows of 128 bytes, 16-byte chunks XOR-swizzled with (row >> 1) & 7 on the global side
    // and in the fragment reads: conflict-free b128 reads), four 36 KB stages with three k-tiles in flight (the y1 / W2 regions are not
    // live yet), and the fragments of k-tile t + 1 are read into a second register set right after the barrier that publishes them,
    // under the second k-step of tile t.  Same products in the same order: bit-identical.
    constexpr int NS = 4, XR = K::NPIX1, SROWS = XR + MID, STG = SROWS * 128, NBLK = SROWS / 8, TDMA = (NBLK + NW - 1) / NW;
    static_assert(K::NK1 % NS == 0 && NS * STG <= K::LDS && SROWS % 8 == 0 && XR % 8 == 0 && MTW1 == 3 && NTW1 == 1, "phase-1 DMA stages");
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1), 0, (unsigned)(MID * CIN * 4), 0x00020000);
    unsigned go[TDMA];                                       // per DMA instruction of this wave: global byte offset of its lane's 16-byte chunk in k-tile 0
    int gblk[TDMA];                                          // ... and its 8-row block of the stage (wave-uniform); blocks < XR / 8 hold x rows
#pragma unroll
    for (int u = 0; u < TDMA; u++) {
      int blk = wv + NW * u;
      if (blk >= NBLK) blk = wv;                             // (36 blocks over 8 waves: the last four waves fetch their first block twice -- keeps vmcnt uniform)
      gblk[u] = blk;
      const int r = blk * 8 + (lane >> 3), lc = (lane & 7) ^ ((r >> 1) & 7);
      if (blk < XR / 8) {
        const int pix = (ty0 - 1) * IW + r;                  // rows above / below the image: out-of-range offset -> zeros
        go[u] = (pix >= 0 && pix < p.H * IW) ? (unsigned)((pix * CIN + lc * 4) * 4) : 0x80000000u;
      } else {
        go[u] = (unsigned)(((r - XR) * CIN + lc * 4) * 4);
      }
    }
#define SSG_BN_DMA1(T_, ST_)                                                                                         \
    { _Pragma("unroll") for (int u = 0; u < TDMA; u++)                                                               \
        bn_dma16(gblk[u] < XR / 8 ? xrsrc : wrsrc, (unsigned)(uintptr_t)(smem + (ST_) * STG + gblk[u] * 1024), go[u], (T_) * 128); }
    // fragments: lane (row l32 of a 32-row MFMA tile, channel group h of a 16-channel k-step): hi = logical chunk 4 ks + 2 h, lo = the next one
    // (the sixth pixel tile of the second wave row does not exist: its fragment reads land in the W1 rows of the stage, its products are
    // never written -- cheaper than a branch around every MFMA)
    const int i1s = __builtin_amdgcn_readfirstlane(i1b);
    int xo[MTW1], xs[MTW1];
#pragma unroll
    for (int i = 0; i < MTW1; i++) { const int r = (i1s + i) * 32 + l32; xo[i] = r * 128; xs[i] = (r >> 1) & 7; }
    const int wr_ = XR + j1b * 32 + l32, wo = wr_ * 128, ws = (wr_ >> 1) & 7;
    v8h fxh[2][2][MTW1], fxl[2][2][MTW1], fwh[2][2], fwl[2][2];       // [register set][k-step][tile]
#define SSG_BN_READS1(ST_, S_)                                                                                       \
    { const unsigned char* sb_ = smem + (ST_) * STG;                                                                 \
      _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                             \
        _Pragma("unroll") for (int i = 0; i < MTW1; i++) {                                        \
          fxh[S_][ks][i] = *reinterpret_cast<const v8h*>(sb_ + xo[i] + (((4 * ks + 2 * h) ^ xs[i]) * 16));           \
          fxl[S_][ks][i] = *reinterpret_cast<const v8h*>(sb_ + xo[i] + (((4 * ks + 2 * h + 1) ^ xs[i]) * 16)); }     \
        fwh[S_][ks] = *reinterpret_cast<const v8h*>(sb_ + wo + (((4 * ks + 2 * h) ^ ws) * 16));                       \
        fwl[S_][ks] = *reinterpret_cast<const v8h*>(sb_ + wo + (((4 * ks + 2 * h + 1) ^ ws) * 16)); } }
#define SSG_BN_MMAK(S_, KS_)                                                                                         \
    { _Pragma("unroll") for (int i = 0; i < MTW1; i++) acc1[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[S_][KS_], fxl[S_][KS_][i], acc1[i][0], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < MTW1; i++) acc1[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[S_][KS_], fxh[S_][KS_][i], acc1[i][0], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < MTW1; i++) acc1[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[S_][KS_], fxh[S_][KS_][i], acc1[i][0], 0, 0, 0); }
    // k-tile t (fragments in set S): first k-step, publish tile t + 1 (and: everybody is done reading this stage), refill this stage with
    // tile t + NS, read tile t + 1's fragments into the other set, second k-step
#define SSG_BN_STEPD(KT_, ST_, STN_, S_)                                                                             \
    { SSG_BN_MMAK(S_, 0)                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TDMA * (NS - 2)) : "memory");                  \
      { const int tn_ = (KT_) + NS < K::NK1 ? (KT_) + NS : K::NK1 - 1;   /* the tail re-fetches the last tile: uniform vmcnt accounting */ \
        SSG_BN_DMA1(tn_, ST_) }                                                                                      \
      SSG_BN_READS1(STN_, 1 - (S_))                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      SSG_BN_MMAK(S_, 1) }
    SSG_BN_DMA1(0, 0) SSG_BN_DMA1(1, 1) SSG_BN_DMA1(2, 2) SSG_BN_DMA1(3, 3)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TDMA * (NS - 1)) : "memory");       // k-tile 0 landed for everybody
    SSG_BN_READS1(0, 0)
#pragma unroll 1
    for (int kt = 0; kt < K::NK1; kt += NS) {
      SSG_BN_STEPD(kt, 0, 1, 0) SSG_BN_STEPD(kt + 1, 1, 2, 1) SSG_BN_STEPD(kt + 2, 2, 3, 0) SSG_BN_STEPD(kt + 3, 3, 0, 1)
    }
    __syncthreads();                       // drains the redundant tail DMAs: the stages are about to become y1 / the W2 stages
#undef SSG_BN_STEPD
#undef SSG_BN_MMAK
#undef SSG_BN_READS1
#undef SSG_BN_DMA1
  } else {
  unsigned aoff[AU];
#pragma unroll
  for (int u = 0; u < AU; u++) {
    const int hp = r1 + RPP1 * u, pix = (ty0 - 1) * IW + hp;  // rows above / below the image (and the padding rows of the stage): out-of-range offset -> the load returns zeros
    aoff[u] = (hp < K::NPIX1 && pix >= 0 && pix < p.H * IW) ? (unsigned)((pix * CIN + ck1 * 4) * 4) : 0x80000000u;
  }
  const float* w1p = p.w1 + (int64_t)r1 * CIN + ck1 * 4;
  v4f sa[K::PD1][AU], sw[K::PD1][WU];
#define SSG_BN_LOAD1(T_, S_)                                                                                         \
  {                                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < AU; u++) {                                                                  \
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[u], (T_) * (K::K1T * 4), 0);                  \
      sa[S_][u] = __builtin_bit_cast(v4f, raw);                                                                     \
    }                                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < WU; u++) sw[S_][u] = *reinterpret_cast<const v4f*>(w1p + (int64_t)(RPP1 * u) * CIN + (T_) * K::K1T); \
  }
#define SSG_BN_STORE1(BUF_, S_)                                                                                      \
  {                                                                                                                  \
    unsigned char* sb_ = smem + (BUF_) * K::BUF1;                                                                    \
    _Pragma("unroll") for (int u = 0; u < AU; u++) *reinterpret_cast<v4f*>(sb_ + (r1 + RPP1 * u) * K::P1T + ck1 * 16) = sa[S_][u]; \
    _Pragma("unroll") for (int u = 0; u < WU; u++) *reinterpret_cast<v4f*>(sb_ + (K::XROWS + r1 + RPP1 * u) * K::P1T + ck1 * 16) = sw[S_][u]; \
  }
#define SSG_BN_MMA1(BUF_)                                                                                            \
  {                                                                                                                  \
    const unsigned char* sb_ = smem + (BUF_) * K::BUF1;                                                              \
    _Pragma("unroll") for (int ks = 0; ks < K::KS1; ks++) {                                                          \
    v8h xh_[MTW1], xl_[MTW1], wh_[NTW1], wl_[NTW1];                                                                  \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) {                                                               \
      const unsigned char* q_ = sb_ + ((i1b + i) * 32 + l32) * K::P1T + ks * 64 + h * 32;                            \
      xh_[i] = *reinterpret_cast<const v8h*>(q_); xl_[i] = *reinterpret_cast<const v8h*>(q_ + 16); }                 \
    _Pragma("unroll") for (int j = 0; j < NTW1; j++) {                                                               \
      const unsigned char* q_ = sb_ + (K::XROWS + (j1b + j) * 32 + l32) * K::P1T + ks * 64 + h * 32;                 \
      wh_[j] = *reinterpret_cast<const v8h*>(q_); wl_[j] = *reinterpret_cast<const v8h*>(q_ + 16); }                 \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[j], xl_[i], acc1[i][j], 0, 0, 0);                      \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[j], xh_[i], acc1[i][j], 0, 0, 0);                      \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[j], xh_[i], acc1[i][j], 0, 0, 0);                      \
    }                                                                                                                \
  }
  // the register set of a tile is a LITERAL index (kt % PD1 of an unrolled loop variable leaves the rings in scratch memory)
#define SSG_BN_STEP1(KT_, S_)                                                                                        \
  {                                                                                                                  \
    if ((KT_) + K::PD1 < K::NK1) SSG_BN_LOAD1((KT_) + K::PD1, S_)   /* the set of tile KT_ went to LDS one step ago */ \
    SSG_BN_MMA1((S_) & 1)                                                                                            \
    if ((KT_) + 1 < K::NK1) SSG_BN_STORE1(((S_) + 1) & 1, ((S_) + 1) % K::PD1)                                        \
    __syncthreads();                                                                                                 \
  }
  static_assert(K::NK1 % K::PD1 == 0 && K::PD1 <= 8, "phase-1 ring of PD1 register sets");
#define SSG_BN_LOAD1_IF(S_) if constexpr ((S_) < K::PD1) SSG_BN_LOAD1(S_, S_)
  SSG_BN_LOAD1_IF(0) SSG_BN_LOAD1_IF(1) SSG_BN_LOAD1_IF(2) SSG_BN_LOAD1_IF(3) SSG_BN_LOAD1_IF(4) SSG_BN_LOAD1_IF(5) SSG_BN_LOAD1_IF(6) SSG_BN_LOAD1_IF(7)
#undef SSG_BN_LOAD1_IF
  SSG_BN_STORE1(0, 0)
  __syncthreads();
#define SSG_BN_STEP1_IF(KT_, S_) if constexpr ((S_) < K::PD1) SSG_BN_STEP1(KT_, S_)
#pragma unroll
  for (int kt0 = 0; kt0 < K::NK1; kt0 += K::PD1) {
    SSG_BN_STEP1_IF(kt0, 0) SSG_BN_STEP1_IF(kt0 + 1, 1) SSG_BN_STEP1_IF(kt0 + 2, 2) SSG_BN_STEP1_IF(kt0 + 3, 3)
    SSG_BN_STEP1_IF(kt0 + 4, 4) SSG_BN_STEP1_IF(kt0 + 5, 5) SSG_BN_STEP1_IF(kt0 + 6, 6) SSG_BN_STEP1_IF(kt0 + 7, 7)
  }
#undef SSG_BN_STEP1_IF
#undef SSG_BN_STEP1
  }
  SSG_BN_STAMP(1)
#undef SSG_BN_LOAD1
#undef SSG_BN_STORE1
#undef SSG_BN_MMA1

  // ---- conv2 weights: first k-tiles on their way while y1 is written
  constexpr int CPR2 = 8 * K::KT2, RP8 = K::NTHR / CPR2, BU = MID / RP8;   // 16-byte pieces per thread and W2 stage (MID rows x KT2 x 128 B)
  const int ck8 = tid % CPR2, r8 = tid / CPR2;
  const float* w2p = p.w2 + (int64_t)r8 * (9 * MID) + ck8 * 4;
  v4f sb2[K::PD2][BU];
#define SSG_BN_LOAD2(T_, S_)                                                                                         \
  { _Pragma("unroll") for (int u = 0; u < BU; u++) sb2[S_][u] = *reinterpret_cast<const v4f*>(w2p + (int64_t)(RP8 * u) * (9 * MID) + (T_) * (32 * K::KT2)); }
#define SSG_BN_STORE2(BUF_, S_)                                                                                      \
  { unsigned char* sb_ = smem + K::W2_OFF + (BUF_) * K::BUF2;                                                        \
    _Pragma("unroll") for (int u = 0; u < BU; u++) *reinterpret_cast<v4f*>(sb_ + (r8 + RP8 * u) * K::P2 + ck8 * 16) = sb2[S_][u]; }
#define SSG_BN_LOAD2_IF(S_) if constexpr ((S_) < K::PD2) SSG_BN_LOAD2(S_, S_)
  SSG_BN_LOAD2_IF(0) SSG_BN_LOAD2_IF(1) SSG_BN_LOAD2_IF(2) SSG_BN_LOAD2_IF(3) SSG_BN_LOAD2_IF(4) SSG_BN_LOAD2_IF(5)
#undef SSG_BN_LOAD2_IF

  // ---- y1 -> LDS (h8l8 pixel rows).  Rows outside the image are conv2's zero padding, not relu(bias).
  unsigned ovf = 0u;
#pragma unroll
  for (int i = 0; i < MTW1; i++) {
    if (i1b + i >= MT1) continue;                           // (a padding tile of the last wave row)
    const int hp0 = (i1b + i) * 32;                         // first halo pixel of this MFMA tile
    const int irow = ty0 - 1 + (hp0 + l32) / IW;
    const bool inside = irow >= 0 && irow < p.H;
#pragma unroll
    for (int j = 0; j < NTW1; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ch = (j1b + j) * 32 + 8 * q + 4 * h;
        const float4 cs = cs1r[j][q], bi = b1r[j][q];
        float4 v = make_float4(acc1[i][j][4 * q] * cs.x + bi.x, acc1[i][j][4 * q + 1] * cs.y + bi.y, acc1[i][j][4 * q + 2] * cs.z + bi.z,
                               acc1[i][j][4 * q + 3] * cs.w + bi.w);
        v = relu4(v);
        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 hi, lo;
        encode4(v, hi, lo);
        ovf |= hi_nonfinite_bits(hi);
        unsigned char* d = smem + (hp0 + l32) * K::PY + (ch >> 3) * 32 + h * 8;
        *reinterpret_cast<uint2*>(d) = hi; *reinterpret_cast<uint2*>(d + 16) = lo;
      }
  }
  if (!K::ZERO_EARLY && tid < K::PY / 16) *reinterpret_cast<uint4*>(smem + K::ZERO_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);   // (phase 1 is over: its stages covered this row)
  SSG_BN_STORE2(0, 0)
  __syncthreads();
  SSG_BN_STAMP(2)

  // =========================== phase 2: y2 = relu(conv2_3x3(y1)), pixel operand from LDS ===========================
  // waves: 4 pixel tiles x WC2 channel groups of NT2 channel tiles each
  constexpr int WC2 = NW / 4, NT2 = MID / 32 / WC2;
  const int pt2 = wave / WC2, cb2 = (wave % WC2) * NT2;
  v16f acc2[NT2];
#pragma unroll
  for (int j = 0; j < NT2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc2[j][r] = 0.f;
  const int m2 = pt2 * 32 + l32, ty2 = m2 / IW, tx2 = m2 - ty2 * IW;       // this lane's output pixel (tile-local)
  float4 cs2r[NT2][4], b2r[NT2][4];                        // needed right after the loop
#pragma unroll
  for (int j = 0; j < NT2; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      cs2r[j][q] = *reinterpret_cast<const float4*>(p.cs2 + (cb2 + j) * 32 + 8 * q + 4 * h); b2r[j][q] = *reinterpret_cast<const float4*>(p.b2 + (cb2 + j) * 32 + 8 * q + 4 * h);
    }
#define SSG_BN_STEP2(KT_, S_, B_)                                                                                     \
  {                                                                                                                  \
    if ((KT_) + K::PD2 < K::NK2) SSG_BN_LOAD2((KT_) + K::PD2, S_)                                                     \
    _Pragma("unroll") for (int sub = 0; sub < K::KT2; sub++) {                                                       \
      const int kt_ = (KT_) * K::KT2 + sub;                                                                          \
      const int chunk = kt_ / 9, tap = kt_ - chunk * 9, r = tap / 3, s = tap - r * 3;                                \
      const int xin = tx2 + s - 1;                                                                                   \
      const int abase = (xin >= 0 && xin < IW) ? ((ty2 + r) * IW + xin) * K::PY : K::ZERO_OFF;                       \
      const unsigned char* wb = smem + K::W2_OFF + (B_) * K::BUF2 + (cb2 * 32 + l32) * K::P2 + sub * 128 + h * 32;     \
      v8h xh_[2], xl_[2], wh_[2][NT2], wl_[2][NT2];                                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                             \
        const unsigned char* q_ = smem + abase + (chunk * 4 + ks * 2 + h) * 32;                                      \
        xh_[ks] = *reinterpret_cast<const v8h*>(q_); xl_[ks] = *reinterpret_cast<const v8h*>(q_ + 16);               \
        _Pragma("unroll") for (int j = 0; j < NT2; j++) {                                                            \
          wh_[ks][j] = *reinterpret_cast<const v8h*>(wb + j * 32 * K::P2 + ks * 64);                                 \
          wl_[ks][j] = *reinterpret_cast<const v8h*>(wb + j * 32 * K::P2 + ks * 64 + 16); }                          \
      }                                                                                                              \
      _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                             \
        _Pragma("unroll") for (int j = 0; j < NT2; j++) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[ks][j], xl_[ks], acc2[j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NT2; j++) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[ks][j], xh_[ks], acc2[j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NT2; j++) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[ks][j], xh_[ks], acc2[j], 0, 0, 0); \
      }                                                                                                              \
    }                                                                                                                \
    if ((KT_) + 1 < K::NK2) SSG_BN_STORE2((B_) ^ 1, ((S_) + 1) % K::PD2)                                              \
    __syncthreads();                                                                                                 \
  }
  // six steps per trip: the register set of step i is i % PD2 (PD2 = 6 or 3), its LDS buffer i & 1
  static_assert(K::NK2 % 6 == 0 && 6 % K::PD2 == 0, "phase-2 ring");
#pragma unroll
  for (int kt0 = 0; kt0 < K::NK2; kt0 += 6) {
    SSG_BN_STEP2(kt0, 0 % K::PD2, 0) SSG_BN_STEP2(kt0 + 1, 1 % K::PD2, 1) SSG_BN_STEP2(kt0 + 2, 2 % K::PD2, 0)
    SSG_BN_STEP2(kt0 + 3, 3 % K::PD2, 1) SSG_BN_STEP2(kt0 + 4, 4 % K::PD2, 0) SSG_BN_STEP2(kt0 + 5, 5 % K::PD2, 1)
  }
#undef SSG_BN_STEP2
  SSG_BN_STAMP(3)
#undef SSG_BN_LOAD2
#undef SSG_BN_STORE2

  // ---- conv3 weights: k-tiles 0 and 1 on their way while y2 is written (every wave is past its last y1 / W2 read)
  constexpr int CU3 = C / RPP;                              // 16-byte pieces per thread and W3 k-tile (C rows x 64 B)
  constexpr int K3 = MID + (DS ? CIN : 0);                 // conv3's reduction length (the weight row)
  const float* w3p = p.w3 + (int64_t)r0 * K3 + ck * 4;
  v4f sc3[2][CU3];
#define SSG_BN_LOAD3(T_, S_)                                                                                         \
  { _Pragma("unroll") for (int u = 0; u < CU3; u++) sc3[S_][u] = *reinterpret_cast<const v4f*>(w3p + (int64_t)(RPP * u) * K3 + (T_) * 16); }
#define SSG_BN_STORE3(BUF_, S_)                                                                                      \
  { unsigned char* sb_ = smem + K::W3_OFF + (BUF_) * K::BUF3;                                                        \
    _Pragma("unroll") for (int u = 0; u < CU3; u++) *reinterpret_cast<v4f*>(sb_ + (r0 + RPP * u) * K::P1 + ck * 16) = sc3[S_][u]; }
  SSG_BN_LOAD3(0, 0)
  SSG_BN_LOAD3(1, 1)
  // phase-3 waves: 4 pixel tiles x WC3 channel groups of NT3 channel tiles each
  constexpr int WC3 = NW / 4, NT3 = C / 32 / WC3;
  const int pt3 = wave / WC3, cb3 = (wave % WC3) * NT3;
  const int64_t gpix0 = ((int64_t)img * p.H + ty0) * IW + pt3 * 32;        // first output pixel of this wave (pixels are contiguous)
  // DS: the downsample operand, this lane's pixel, k-tiles MID/16 .. NK3-1 = channels of x: [8 hi][8 lo] of group h per k-tile
  constexpr int NXK = DS ? CIN / 16 : 1;
  v8h xrh[NXK], xrl[NXK];
  if constexpr (DS) {
    const float* xq = p.x + (gpix0 + l32) * CIN + h * 8;
#pragma unroll
    for (int t = 0; t < NXK; t++) { xrh[t] = *reinterpret_cast<const v8h*>(xq + t * 16); xrl[t] = *reinterpret_cast<const v8h*>(xq + t * 16 + 4); }
  }
  // y2 rows: phase 2 writes the rows of pixel tile pt2 (its channel group), phase 3 reads those of pt3 (4 waves: the same
  // wave, 8 waves: a wave pair -- the barrier before the first multiply of phase 3 publishes them either way)
  unsigned char* myrows = smem + (pt3 * 32) * K::PY;
#pragma unroll
  for (int j = 0; j < NT2; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ch = (cb2 + j) * 32 + 8 * q + 4 * h;
      const float4 cs = cs2r[j][q], bi = b2r[j][q];
      float4 v = make_float4(acc2[j][4 * q] * cs.x + bi.x, acc2[j][4 * q + 1] * cs.y + bi.y, acc2[j][4 * q + 2] * cs.z + bi.z, acc2[j][4 * q + 3] * cs.w + bi.w);
      v = relu4(v);
      uint2 hi, lo;
      encode4(v, hi, lo);
      ovf |= hi_nonfinite_bits(hi);
      unsigned char* d = smem + (pt2 * 32 + l32) * K::PY + (ch >> 3) * 32 + h * 8;
      *reinterpret_cast<uint2*>(d) = hi; *reinterpret_cast<uint2*>(d + 16) = lo;
    }

  // =========================== phase 3: out = relu(conv3(y2) + x) ===========================
  static_assert(K::NK3 % 2 == 0, "phase 3 takes its k-tiles in rounds of two buffers");
  v16f acc3[NT3];
#pragma unroll
  for (int j = 0; j < NT3; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc3[j][r] = 0.f;
#define SSG_BN_MMA3K(BUF_, KT_)                                                                                      \
  {                                                                                                                  \
    v8h xh_, xl_;                                                                                                    \
    if constexpr (DS && (KT_) >= MID / 16) { xh_ = xrh[((KT_) - MID / 16) % NXK]; xl_ = xrl[((KT_) - MID / 16) % NXK]; } \
    else {                                                                                                           \
      const unsigned char* q_ = myrows + l32 * K::PY + (((KT_) % (MID / 16)) * 2 + h) * 32;                          \
      xh_ = *reinterpret_cast<const v8h*>(q_); xl_ = *reinterpret_cast<const v8h*>(q_ + 16);                         \
    }                                                                                                                \
    const unsigned char* wb_ = smem + K::W3_OFF + (BUF_) * K::BUF3 + (cb3 * 32 + l32) * K::P1 + h * 32;              \
    _Pragma("unroll") for (int jj = 0; jj < NT3; jj += 4) {                                                          \
      v8h wh_[4], wl_[4];                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                \
        wh_[j] = *reinterpret_cast<const v8h*>(wb_ + (jj + j) * 32 * K::P1); wl_[j] = *reinterpret_cast<const v8h*>(wb_ + (jj + j) * 32 * K::P1 + 16); } \
      _Pragma("unroll") for (int j = 0; j < 4; j++) acc3[jj + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[j], xl_, acc3[jj + j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < 4; j++) acc3[jj + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[j], xh_, acc3[jj + j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < 4; j++) acc3[jj + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[j], xh_, acc3[jj + j], 0, 0, 0); \
    }                                                                                                                \
  }
  // epilogue addressing / operands (the first channel tile's are fetched during the last round of the multiply)
  constexpr int EP = 36, CPR = 8, RPI = 8, ITS = 4;
  const int chunk = lane % CPR, prow = lane / CPR, odd = lane & 1;
  const float* __restrict__ resp = p.x + gpix0 * C;        // residual (identity blocks only)
  float* __restrict__ outp = p.out + gpix0 * C;
  float4 rr[2][ITS], b3r[2], cs3r[2];
#ifdef SSG_BN_ABL_NORES                                  // ablation build (tools/micro/bneck_prof.hip): no residual read, wrong results
#define SSG_BN_RESLOAD(P_) make_float4(0.f, 0.f, 0.f, 0.f)
#elif SSG_BN_NT_RES                                      // the residual re-read is x's last use: nt cache policy (see SSG_BN_NT_STORE above)
#define SSG_BN_RESLOAD(P_) ([&]() { const v4f t_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(P_)); return make_float4(t_[0], t_[1], t_[2], t_[3]); }())
#else
#define SSG_BN_RESLOAD(P_) (*reinterpret_cast<const float4*>(P_))
#endif
  SSG_BN_STORE3(0, 0)
  SSG_BN_STORE3(1, 1)
  if constexpr (K::NK3 > 2) { SSG_BN_LOAD3(2, 0) SSG_BN_LOAD3(3, 1) }
  __syncthreads();
  static_assert(K::NK3 / 2 <= 4, "at most four rounds");
#define SSG_BN_ROUND3(R_)                                                                                            \
  if constexpr ((R_) < K::NK3 / 2) {                                                                                 \
    SSG_BN_MMA3K(0, 2 * (R_))                                                                                        \
    SSG_BN_MMA3K(1, 2 * (R_) + 1)                                                                                    \
    if constexpr ((R_) + 1 < K::NK3 / 2) {                                                                           \
      __syncthreads();                       /* everybody is done reading the two buffers */                         \
      SSG_BN_STORE3(0, 0)                                                                                            \
      SSG_BN_STORE3(1, 1)                                                                                            \
      if constexpr ((R_) + 2 < K::NK3 / 2) { SSG_BN_LOAD3(2 * (R_) + 4, 0) SSG_BN_LOAD3(2 * (R_) + 5, 1) }           \
      else {                                 /* last round ahead: the W3 staging registers are free for the epilogue's first operands */ \
        if constexpr (!DS) {                                                                                         \
          _Pragma("unroll") for (int it = 0; it < ITS; it++) rr[0][it] = SSG_BN_RESLOAD(resp + (int64_t)(it * RPI + prow) * C + cb3 * 32 + chunk * 4); \
        }                                                                                                            \
        b3r[0] = *reinterpret_cast<const float4*>(p.b3 + cb3 * 32 + chunk * 4); cs3r[0] = *reinterpret_cast<const float4*>(p.cs3 + cb3 * 32 + chunk * 4); \
      }                                                                                                              \
      __syncthreads();                                                                                               \
    }                                                                                                                \
  }
  SSG_BN_ROUND3(0) SSG_BN_ROUND3(1) SSG_BN_ROUND3(2) SSG_BN_ROUND3(3)
#undef SSG_BN_ROUND3
  SSG_BN_STAMP(4)
#undef SSG_BN_LOAD3
#undef SSG_BN_STORE3
#undef SSG_BN_MMA3K

  // ---- epilogue: per channel tile a 32-pixel x 32-channel patch through LDS (this wave's own, now dead, y2 rows), then
  // whole 128-byte row segments: bias, residual (x, h8l8), ReLU, re-encode, store.  Same code path as conv.hip.
  static_assert(32 * EP * 4 <= 32 * K::PY, "patch fits in the wave's y2 rows");
  float* patch;
  if constexpr (WC3 == 1) patch = reinterpret_cast<float*>(myrows);       // its y2 rows are this wave's alone and dead now
  else {                                                                  // rows shared by a wave pair: the patches take over the W3 stages
    __syncthreads();
    patch = reinterpret_cast<float*>(smem + K::W3_OFF) + wave * (32 * EP);
  }
#pragma unroll
  for (int j = 0; j < NT3; j++) {
    const int col = (cb3 + j) * 32 + chunk * 4;
    if (j + 1 < NT3) {
      if constexpr (!DS) {
#pragma unroll
        for (int it = 0; it < ITS; it++) rr[(j + 1) & 1][it] = SSG_BN_RESLOAD(resp + (int64_t)(it * RPI + prow) * C + col + 32);
      }
      b3r[(j + 1) & 1] = *reinterpret_cast<const float4*>(p.b3 + col + 32); cs3r[(j + 1) & 1] = *reinterpret_cast<const float4*>(p.cs3 + col + 32);
    }
    const float4 bias = b3r[j & 1], cs = cs3r[j & 1];
#pragma unroll
    for (int q = 0; q < 4; q++)
      *reinterpret_cast<float4*>(patch + l32 * EP + 8 * q + 4 * h) = make_float4(acc3[j][4 * q], acc3[j][4 * q + 1], acc3[j][4 * q + 2], acc3[j][4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int it = 0; it < ITS; it++) {
      const int pr = it * RPI + prow;
      float4 v = *reinterpret_cast<const float4*>(patch + pr * EP + chunk * 4);
      v.x = v.x * cs.x + bias.x; v.y = v.y * cs.y + bias.y; v.z = v.z * cs.z + bias.z; v.w = v.w * cs.w + bias.w;
      if constexpr (!DS) {
      const float4 rw = rr[j & 1][it];
      float4 r4;
      {   // even lane holds hi0..7 of the 8-channel group, odd lane lo0..7; each needs hi and lo of ITS four channels
        // (component-wise selects: a select between uint2 / float4 aggregates goes through scratch memory)
        const unsigned a0 = __float_as_uint(rw.x), a1 = __float_as_uint(rw.y), a2 = __float_as_uint(rw.z), a3 = __float_as_uint(rw.w);
        const unsigned g0 = lane_xor1((odd ? a0 : a2)), g1 = lane_xor1((odd ? a1 : a3));
        const unsigned h0 = odd ? g0 : a0, h1 = odd ? g1 : a1, l0 = odd ? a2 : g0, l1 = odd ? a3 : g1;
        r4 = decode4(make_uint2(h0, h1), make_uint2(l0, l1));
      }
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      v = relu4(v);
      uint2 hp, lp;
      encode4(v, hp, lp);
      ovf |= hi_nonfinite_bits(hp);
      const unsigned rx = lane_xor1((odd ? hp.x : lp.x)), ry = lane_xor1((odd ? hp.y : lp.y));
      const uint4 stv = make_uint4(odd ? rx : hp.x, odd ? ry : hp.y, odd ? lp.x : rx, odd ? lp.y : ry);
#if SSG_BN_NT_STORE
      { const v4u sv_ = {stv.x, stv.y, stv.z, stv.w}; __builtin_nontemporal_store(sv_, reinterpret_cast<v4u*>(outp + (int64_t)pr * C + col)); }
#else
      *reinterpret_cast<uint4*>(outp + (int64_t)pr * C + col) = stv;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if ((ovf & 0x80008000u) && p.overflow) *p.overflow = 1;
  SSG_BN_STAMP(5)
}

}  // namespace bneck
}  // namespace ssg

// 1 when ssg_bottleneck_nhwc_x / ssg_bottleneck_ds_nhwc_x has a kernel for this block shape (CIN == C: identity block):
//   layer1 of ResNet-50 at 256x128 input: H x 32 x 256, MID 64 (identity blocks and the first block, CIN = 64), 4-row tiles;
//   layer2 identity blocks: H x 16 x 512, MID 128, 8-row tiles (one 8-wave workgroup per CU: the 128-channel halo intermediate is 85 KB)
extern "C" int ssg_bottleneck_supported(int H, int W, int CIN, int C, int MID) {
  if (C == 256 && MID == 64 && (CIN == 256 || CIN == 64) && W == 32 && H > 0 && H % 4 == 0) return 1;
  if (C == 512 && MID == 128 && CIN == 512 && W == 16 && H > 0 && H % 8 == 0) return 1;
  return 0;
}

template <int C, int MID, int IW, int TH, int CIN, int NW>
static int launch_bottleneck(const ssg::bneck::Params& p, hipStream_t stream) {
  using namespace ssg::bneck;
  using K = Cfg<C, MID, IW, TH, CIN, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    SSG_HIP(hipFuncSetAttribute((const void*)bottleneck_kernel<C, MID, IW, TH, CIN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((bottleneck_kernel<C, MID, IW, TH, CIN, NW>), dim3(p.B * (p.H / TH)), dim3(NW * 64), K::LDS, stream, p);
  SSG_LAUNCH_CHECK("bottleneck_kernel");
  return SSG_OK;
}

// Identity bottleneck block (no downsample branch, stride 1), split-half tensors:
//   out = relu(conv3(relu(conv2(relu(conv1(x))))) + x),  x / out [B,H,W,C] h8l8, weights as ssg_conv2d_nhwc_x takes them
//   (w1 [MID][C], w2 [MID][9*MID] in the k = (32-channel chunk, tap, channel) order, w3 [C][MID], each row pre-multiplied by a
//   power of two that ch_scale* undoes), biases fp32.  out must not alias x (halo rows are read by other workgroups).
extern "C" int ssg_bottleneck_nhwc_x(const void* x, const void* w1, const float* b1, const float* cs1, const void* w2, const float* b2, const float* cs2,
                                     const void* w3, const float* b3, const float* cs3, void* out, int B, int H, int W, int C, int MID,
                                     int32_t* overflow, hipStream_t stream) {
  using namespace ssg::bneck;
  if (B <= 0 || !ssg_bottleneck_supported(H, W, C, C, MID) || !cs1 || !cs2 || !cs3 || x == out) {
    ssg_set_error("ssg_bottleneck_nhwc_x: unsupported block B=%d H=%d W=%d C=%d MID=%d (see ssg_bottleneck_supported)", B, H, W, C, MID);
    return SSG_ERR_INVALID;
  }
  Params p;
  p.x = (const float*)x; p.out = (float*)out;
  p.w1 = (const float*)w1; p.b1 = b1; p.cs1 = cs1; p.w2 = (const float*)w2; p.b2 = b2; p.cs2 = cs2; p.w3 = (const float*)w3; p.b3 = b3; p.cs3 = cs3;
  p.B = B; p.H = H; p.overflow = overflow;
  if (C == 512) return launch_bottleneck<512, 128, 16, 8, 512, 8>(p, stream);
  return launch_bottleneck<256, 64, 32, 4, 256, 4>(p, stream);
}

// Bottleneck block with a stride-1 downsample branch (the first block of layer1):
//   out = relu(conv3(relu(conv2(relu(conv1(x))))) + downsample(x)),  x [B,H,W,CIN], out [B,H,W,C] h8l8;
//   w3cat [C][MID + CIN] = conv3 | downsample weights concatenated along K (the layout of ssg_conv1x1_dual_nhwc_x), b3 = b3 + b_ds.
extern "C" int ssg_bottleneck_ds_nhwc_x(const void* x, const void* w1, const float* b1, const float* cs1, const void* w2, const float* b2, const float* cs2,
                                        const void* w3cat, const float* b3, const float* cs3, void* out, int B, int H, int W, int CIN, int C, int MID,
                                        int32_t* overflow, hipStream_t stream) {
  using namespace ssg::bneck;
  if (B <= 0 || CIN == C || !ssg_bottleneck_supported(H, W, CIN, C, MID) || !cs1 || !cs2 || !cs3) {
    ssg_set_error("ssg_bottleneck_ds_nhwc_x: unsupported block B=%d H=%d W=%d CIN=%d C=%d MID=%d (see ssg_bottleneck_supported)", B, H, W, CIN, C, MID);
    return SSG_ERR_INVALID;
  }
  Params p;
  p.x = (const float*)x; p.out = (float*)out;
  p.w1 = (const float*)w1; p.b1 = b1; p.cs1 = cs1; p.w2 = (const float*)w2; p.b2 = b2; p.cs2 = cs2; p.w3 = (const float*)w3cat; p.b3 = b3; p.cs3 = cs3;
  p.B = B; p.H = H; p.overflow = overflow;
  return launch_bottleneck<256, 64, 32, 4, 64, 4>(p, stream);
}
