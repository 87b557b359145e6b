// bottleneck2.hip -- the identity bottleneck blocks of layer2 (H x 16 x 512, mid 128) in ONE kernel, second generation.
//
// reid/models/base.py:57-90 (torchvision Bottleneck without a downsample branch), eval mode, BatchNorm folded:
//   out = relu( conv3_1x1( relu( conv2_3x3( relu( conv1_1x1(x) ) ) ) ) + x )
// bottleneck.hip runs this block as one 8-wave workgroup per CU on 8-row tiles: its 128-channel halo intermediate alone takes
// 85 KB of LDS, the weight stages of conv2 / conv3 another 70 / 82 KB, so ONE workgroup fits a CU and its phases -- conv1 (bound
// by the x stream from HBM), conv2 (bound by the matrix pipe), conv3 + epilogue (residual re-read and stores) -- run one after
// the other with nothing to overlap: 24 % MFMA duty and 1.7 TB/s at the same time (round 2: 1.24 ms per block and 1000 images).
//
// Here a workgroup owns FOUR image rows (64 output pixels, 96 halo pixels) and only the pixel operands live in LDS:
//   phase 1  y1[96, 128]  = relu(x W1^T)          x k-tiles (HBM) and W1 k-tiles staged global -> registers -> LDS as in bottleneck.hip
//   phase 2  y2[64, 128]  = relu(conv3x3(y1))     pixel fragments from LDS (a tap = an address offset); a wave owns ONE 32-channel
//                                                 tile for all 64 pixels and loads its W2 fragments straight from L2 into registers
//                                                 (each weight is loaded by exactly one wave of the workgroup: no LDS stage, no barrier)
//   phase 3  out[64, 512] = relu(y2 W3^T + x)     a wave owns 128 output channels for all 64 pixels, W3 fragments straight into registers
// LDS: 52 KB (y1 + a zero pixel row; y2 and the epilogue patches reuse it) -> two 4-wave workgroups per CU (the register budget of
// 256 allows no third): one's x stream runs under the other's multiplies.  The price is the weight traffic from L2 (1.1 MB per
// 64 pixels instead of per 128) and 1.5 instead of 1.25 halo rows per output row in conv1.
//
// Numerics: the same k-steps, the same three-product order per k-step and the same epilogues as bottleneck.hip and as the three
// separate launches of conv.hip -> bit-identical block output (tests/test_gpu_parity.py: fused vs separate launches).
#include "ssg_common.h"

namespace ssg {
namespace bneck2 {

using bneck::Params;
using bneck::v16f;
using bneck::v4f;
using bneck::v4u;
using bneck::v8h;
using bneck::decode4;
using bneck::encode4;
using bneck::hi_nonfinite_bits;
using bneck::relu4;

constexpr int C = 512, MID = 128, IW = 16, TH = 4;
constexpr int NW = 4, NTHR = 256;
constexpr int HROWS = TH + 2, NPIX1 = HROWS * IW, NPIX = TH * IW;        // 96 halo pixels, 64 output pixels
constexpr int RPP = NTHR / 4;                                              // 64-byte k-tile rows staged per pass of the workgroup
constexpr int XROWS = (NPIX1 + RPP - 1) / RPP * RPP;                       // 128: x rows of a phase-1 stage (padding rows load zeros)
constexpr int P1 = 80;                                                     // LDS pitch of a 64-byte k-tile row
constexpr int PY = MID * 4 + 16;                                           // LDS pitch of a y1 / y2 pixel row (h8l8, all 128 channels)
constexpr int BUF1 = (XROWS + MID) * P1;                                   // phase-1 stage: x rows, then W1 rows
constexpr int ZERO_OFF = NPIX1 * PY;                                       // one all-zero pixel row (conv2's left / right padding)
constexpr int PATCH_OFF = NPIX * PY;                                       // epilogue patches behind y2 (y1's tail is dead by then)
constexpr int EP = 36;                                                     // patch pitch in floats
constexpr int LDS = PATCH_OFF + NW * 32 * EP * 4 > ZERO_OFF + PY ? PATCH_OFF + NW * 32 * EP * 4 : ZERO_OFF + PY;
constexpr int NK1 = C / 16, PD1 = 4;                                       // conv1: 32 k-tiles of 16 channels, 4 register sets in flight
constexpr int NKT2 = (MID / 32) * 9;                                       // conv2: 36 (32-channel chunk, tap) k-tiles of two k-steps each
constexpr int PD2 = 6;                                                     // conv2: k-steps of W2 fragments in flight
constexpr int NK3 = MID / 16;                                              // conv3: 8 k-steps
static_assert(2 * BUF1 <= ZERO_OFF && PATCH_OFF + NW * 32 * EP * 4 <= LDS && LDS <= 80 * 1024, "LDS map");
static_assert(NK1 % PD1 == 0 && (2 * NKT2) % PD2 == 0, "register rings");

__global__ __launch_bounds__(NTHR, 2) void bottleneck2_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
  const int tiles_img = p.H / TH, ntiles = p.B * tiles_img;
  int T;
  {   // every XCD gets a contiguous run of tiles: the halo rows of a tile are its neighbours' rows (L2 hits)
    const int b = (int)blockIdx.x, q = ntiles / 8, r = ntiles % 8, x = b % 8, s = b / 8;
    T = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
  }
  const int img = T / tiles_img, ty0 = (T - img * tiles_img) * TH;
  SSG_BN_STAMP(0)
  if (tid < PY / 16) *reinterpret_cast<uint4*>(smem + ZERO_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // =========================== phase 1: y1 = relu(conv1(x)) on the 6 halo rows ===========================
  constexpr int AU = XROWS / RPP, WU = MID / RPP;      // 16-byte pieces per thread and k-tile: x rows, W1 rows (2 + 2)
  const int ck = tid & 3, r0 = tid >> 2;
  const float* ximg = p.x + (int64_t)img * p.H * IW * C;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, (unsigned)(p.H * IW * C * 4), 0x00020000);
  unsigned aoff[AU];
#pragma unroll
  for (int u = 0; u < AU; u++) {
    const int hp = r0 + RPP * u, pix = (ty0 - 1) * IW + hp;      // rows outside the image / padding rows of the stage: out-of-range offset -> zeros
    aoff[u] = (hp < NPIX1 && pix >= 0 && pix < p.H * IW) ? (unsigned)((pix * C + ck * 4) * 4) : 0x80000000u;
  }
  const float* w1p = p.w1 + (int64_t)r0 * C + ck * 4;
  v4f sa[PD1][AU], sw[PD1][WU];
#ifndef SSG_B2_ABL          // ablation switches of tools/micro/bneck2_prof.hip: bit 0 = x loaded for the first k-tiles only, bit 1 = W1 likewise, bit 2 = no phase-1 MFMAs
#define SSG_B2_ABL 0
#endif
#define SSG_B2_LOAD1(T_, S_)                                                                                         \
  {                                                                                                                  \
    if (!(SSG_B2_ABL & 1) || (T_) < PD1) {                                                                            \
    _Pragma("unroll") for (int u = 0; u < AU; u++) {                                                                  \
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[u], (T_) * 64, 0);                            \
      sa[S_][u] = __builtin_bit_cast(v4f, raw);                                                                     \
    } }                                                                                                               \
    if (!(SSG_B2_ABL & 2) || (T_) < PD1) {                                                                            \
    _Pragma("unroll") for (int u = 0; u < WU; u++) sw[S_][u] = *reinterpret_cast<const v4f*>(w1p + (int64_t)(RPP * u) * C + (T_) * 16); } \
  }
#define SSG_B2_STORE1(BUF_, S_)                                                                                      \
  {                                                                                                                  \
    unsigned char* sb_ = smem + (BUF_) * BUF1;                                                                       \
    _Pragma("unroll") for (int u = 0; u < AU; u++) *reinterpret_cast<v4f*>(sb_ + (r0 + RPP * u) * P1 + ck * 16) = sa[S_][u]; \
    _Pragma("unroll") for (int u = 0; u < WU; u++) *reinterpret_cast<v4f*>(sb_ + (XROWS + r0 + RPP * u) * P1 + ck * 16) = sw[S_][u]; \
  }
  // MFMA tiles: 3 pixel tiles (+ one of padding rows) x 4 channel tiles; wave = (pixel tile pair wave / 2, channel tile pair wave % 2)
  constexpr int MT1 = NPIX1 / 32, MTW1 = 2, NTW1 = 2;
  const int i1b = (wave >> 1) * MTW1, j1b = (wave & 1) * NTW1;
  v16f acc1[MTW1][NTW1];
#pragma unroll
  for (int i = 0; i < MTW1; i++)
#pragma unroll
    for (int j = 0; j < NTW1; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[i][j][r] = 0.f;
  // Software pipeline of a step t (fragments of tile t already in registers F[t & 1], tile t + 1 published in LDS buffer (t + 1) & 1):
  //   fragment reads of tile t + 1 -> F[(t + 1) & 1]   |  12 MFMAs on F[t & 1]   |  tile t + 2: registers -> LDS buffer t & 1 (its
  //   tile t was read before the previous barrier), global loads of tile t + 2 + PD1 into the freed register set  |  barrier
  // so the LDS read latency and the LDS writes run under the multiply instead of in front of it (the first version of this phase
  // -- read, multiply, write, barrier, one after the other -- spent 1250 cycles per k-tile in the LDS round trips alone).
  v8h fxh[2][MTW1], fxl[2][MTW1], fwh[2][NTW1], fwl[2][NTW1];
#define SSG_B2_READ1(BUF_, F_)                                                                                       \
  {                                                                                                                  \
    const unsigned char* sb_ = smem + (BUF_) * BUF1;                                                                 \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) {                                                               \
      const unsigned char* q_ = sb_ + ((i1b + i) * 32 + l32) * P1 + h * 32;                                          \
      fxh[F_][i] = *reinterpret_cast<const v8h*>(q_); fxl[F_][i] = *reinterpret_cast<const v8h*>(q_ + 16); }         \
    _Pragma("unroll") for (int j = 0; j < NTW1; j++) {                                                               \
      const unsigned char* q_ = sb_ + (XROWS + (j1b + j) * 32 + l32) * P1 + h * 32;                                  \
      fwh[F_][j] = *reinterpret_cast<const v8h*>(q_); fwl[F_][j] = *reinterpret_cast<const v8h*>(q_ + 16); }         \
  }
#define SSG_B2_MMA1(F_)                                                                                              \
  {                                                                                                                  \
    if (SSG_B2_ABL & 4) {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < MTW1; i++) asm volatile("" ::"v"(fxh[F_][i]), "v"(fxl[F_][i]));           \
      _Pragma("unroll") for (int j = 0; j < NTW1; j++) asm volatile("" ::"v"(fwh[F_][j]), "v"(fwl[F_][j]));           \
    } else {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[F_][j], fxl[F_][i], acc1[i][j], 0, 0, 0);              \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[F_][j], fxh[F_][i], acc1[i][j], 0, 0, 0);              \
    _Pragma("unroll") for (int i = 0; i < MTW1; i++) _Pragma("unroll") for (int j = 0; j < NTW1; j++)                \
      acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[F_][j], fxh[F_][i], acc1[i][j], 0, 0, 0);              \
    }                                                                                                                \
  }
  // (register sets and fragment sets are LITERAL indices: an index computed from a loop variable leaves them in scratch memory)
  // KT_ = tile being multiplied, P_ = KT_ & 1, S2_ = (KT_ + 2) % PD1 = the register set holding tile KT_ + 2
// (every memory instruction of the step is issued BEFORE the MFMA cluster -- a wave issues in order and an MFMA waits for the
// matrix pipe, so LDS reads placed behind the cluster would start only when it has drained; hipcc moves them there if allowed)
#define SSG_B2_STEP1(KT_, P_, S2_)                                                                                   \
  {                                                                                                                  \
    if ((KT_) + 1 < NK1) SSG_B2_READ1((P_) ^ 1, (P_) ^ 1)                                                            \
    if ((KT_) + 2 < NK1) SSG_B2_STORE1(P_, S2_)                                                                       \
    if ((KT_) + 2 + PD1 < NK1) SSG_B2_LOAD1((KT_) + 2 + PD1, S2_)                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    SSG_B2_MMA1(P_)                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __syncthreads();                                                                                                 \
  }
  SSG_B2_LOAD1(0, 0) SSG_B2_LOAD1(1, 1) SSG_B2_LOAD1(2, 2) SSG_B2_LOAD1(3, 3)
  SSG_B2_STORE1(0, 0)
  SSG_B2_LOAD1(4, 0)
  __syncthreads();
  SSG_B2_READ1(0, 0)                                          // tile 0 -> F[0]
  SSG_B2_STORE1(1, 1)                                         // tile 1 -> buffer 1
  SSG_B2_LOAD1(5, 1)
  __syncthreads();
#ifndef SSG_B2_UNROLL1
#define SSG_B2_UNROLL1 1      // 1 = rolled loops (code size: the kernel walks its instruction stream once per tile)
#endif
#pragma unroll SSG_B2_UNROLL1
  for (int kt0 = 0; kt0 < NK1; kt0 += PD1) { SSG_B2_STEP1(kt0, 0, 2) SSG_B2_STEP1(kt0 + 1, 1, 3) SSG_B2_STEP1(kt0 + 2, 0, 0) SSG_B2_STEP1(kt0 + 3, 1, 1) }
  SSG_BN_STAMP(1)
#undef SSG_B2_STEP1
#undef SSG_B2_LOAD1
#undef SSG_B2_STORE1
#undef SSG_B2_MMA1
#undef SSG_B2_READ1
  float4 cs1r[NTW1][4], b1r[NTW1][4];                   // folded BatchNorm scale / bias of this wave's conv1 channels
#pragma unroll
  for (int j = 0; j < NTW1; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      cs1r[j][q] = *reinterpret_cast<const float4*>(p.cs1 + (j1b + j) * 32 + 8 * q + 4 * h); b1r[j][q] = *reinterpret_cast<const float4*>(p.b1 + (j1b + j) * 32 + 8 * q + 4 * h);
    }

  // ---- conv2 weights of this wave's channel tile (32 output channels x all 64 pixels): the first k-steps on their way while y1 is written
  // a k-step = 16 channels of one tap: lane (l32, h) takes the [8 hi | 8 lo] group h of weight row wave*32 + l32
  const float* w2p = p.w2 + (int64_t)(wave * 32 + l32) * (9 * MID) + h * 8;
  v8h rwh[PD2], rwl[PD2];
#define SSG_B2_LOADW2(KS_, S_) { const float* q_ = w2p + (KS_) * 16; rwh[S_] = *reinterpret_cast<const v8h*>(q_); rwl[S_] = *reinterpret_cast<const v8h*>(q_ + 4); }
  SSG_B2_LOADW2(0, 0) SSG_B2_LOADW2(1, 1) SSG_B2_LOADW2(2, 2) SSG_B2_LOADW2(3, 3) SSG_B2_LOADW2(4, 4)        // k-steps 0 .. PD2 - 2

  // ---- y1 -> LDS (h8l8 pixel rows).  Rows outside the image are conv2's zero padding, not relu(bias).
  unsigned ovf = 0u;
#pragma unroll
  for (int i = 0; i < MTW1; i++) {
    if (i1b + i >= MT1) continue;                           // (the padding tile of the second wave row)
    const int hp0 = (i1b + i) * 32;
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
        unsigned char* d = smem + (hp0 + l32) * PY + (ch >> 3) * 32 + h * 8;
        *reinterpret_cast<uint2*>(d) = hi; *reinterpret_cast<uint2*>(d + 16) = lo;
      }
  }
  __syncthreads();
  SSG_BN_STAMP(2)

  // =========================== phase 2: y2 = relu(conv2_3x3(y1)); wave = channel tile `wave` x both pixel tiles ===========================
  v16f acc2[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc2[i][r] = 0.f;
  int ty2[2], tx2[2];
#pragma unroll
  for (int i = 0; i < 2; i++) { const int m = i * 32 + l32; ty2[i] = m / IW; tx2[i] = m - ty2[i] * IW; }
  float4 cs2r[4], b2r[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { cs2r[q] = *reinterpret_cast<const float4*>(p.cs2 + wave * 32 + 8 * q + 4 * h); b2r[q] = *reinterpret_cast<const float4*>(p.b2 + wave * 32 + 8 * q + 4 * h); }
  // pixel fragments of k-step s + 1 are read from LDS while the MFMAs of k-step s run (two fragment sets, literal indices)
  v8h pxh[2][2], pxl[2][2];
#define SSG_B2_READ2(CH_, LS_, F_)        /* k-step LS_ (literal, 0..17) of 32-channel chunk CH_ (run time) */          \
  { constexpr int tap_ = (LS_) >> 1, ks_ = (LS_) & 1, r_ = tap_ / 3, s_ = tap_ - r_ * 3;                              \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                                  \
      const int xin_ = tx2[i] + s_ - 1;                                                                              \
      const int ab_ = (xin_ >= 0 && xin_ < IW) ? ((ty2[i] + r_) * IW + xin_) * PY : ZERO_OFF;                         \
      const unsigned char* q_ = smem + ab_ + ((CH_) * 4 + ks_ * 2 + h) * 32;                                         \
      pxh[F_][i] = *reinterpret_cast<const v8h*>(q_); pxl[F_][i] = *reinterpret_cast<const v8h*>(q_ + 16); } }
  // one k-step: fragments of the next k-step, the W2 fragments PD2 - 1 k-steps ahead (into the slot consumed one k-step ago), 6 MFMAs
#define SSG_B2_KSTEP2(LS_)                                                                                           \
  { constexpr int slot_ = (LS_) % PD2, f_ = (LS_) & 1, nslot_ = ((LS_) + PD2 - 1) % PD2;                              \
    if ((LS_) + 1 < 18) SSG_B2_READ2(chunk, ((LS_) + 1) % 18, f_ ^ 1)                                                 \
    else if (chunk + 1 < MID / 32) SSG_B2_READ2(chunk + 1, 0, f_ ^ 1)                                                 \
    if (chunk * 18 + (LS_) + PD2 - 1 < 2 * NKT2) SSG_B2_LOADW2(chunk * 18 + (LS_) + PD2 - 1, nslot_)                   \
    __builtin_amdgcn_sched_barrier(0);                      /* memory instructions first, then the MFMA cluster (see phase 1) */ \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[slot_], pxl[f_][i], acc2[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwl[slot_], pxh[f_][i], acc2[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[slot_], pxh[f_][i], acc2[i], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); }
  static_assert(18 % PD2 == 0, "ring slots repeat per chunk");
  SSG_B2_READ2(0, 0, 0)
#pragma unroll SSG_B2_UNROLL1
  for (int chunk = 0; chunk < MID / 32; chunk++) {          // 4 chunks of 32 channels x 9 taps x 2 k-steps
    SSG_B2_KSTEP2(0) SSG_B2_KSTEP2(1) SSG_B2_KSTEP2(2) SSG_B2_KSTEP2(3) SSG_B2_KSTEP2(4) SSG_B2_KSTEP2(5)
    SSG_B2_KSTEP2(6) SSG_B2_KSTEP2(7) SSG_B2_KSTEP2(8) SSG_B2_KSTEP2(9) SSG_B2_KSTEP2(10) SSG_B2_KSTEP2(11)
    SSG_B2_KSTEP2(12) SSG_B2_KSTEP2(13) SSG_B2_KSTEP2(14) SSG_B2_KSTEP2(15) SSG_B2_KSTEP2(16) SSG_B2_KSTEP2(17)
  }
#undef SSG_B2_KSTEP2
#undef SSG_B2_READ2
#undef SSG_B2_LOADW2
  SSG_BN_STAMP(3)

  // ---- conv3 weights of this wave's 128 output channels: k-steps 0 and 1 on their way while y2 is written
  const float* w3p = p.w3 + (int64_t)(wave * 128 + l32) * MID + h * 8;
  v8h r3h[2][4], r3l[2][4];
#define SSG_B2_LOADW3(KT_, S_)                                                                                       \
  { _Pragma("unroll") for (int j = 0; j < 4; j++) { const float* q_ = w3p + (int64_t)(j * 32) * MID + (KT_) * 16;     \
      r3h[S_][j] = *reinterpret_cast<const v8h*>(q_); r3l[S_][j] = *reinterpret_cast<const v8h*>(q_ + 4); } }
  SSG_B2_LOADW3(0, 0)
  SSG_B2_LOADW3(1, 1)
  __syncthreads();                                          // every wave is past its last y1 read: y2 takes over the first 64 pixel rows
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ch = wave * 32 + 8 * q + 4 * h;
      const float4 cs = cs2r[q], bi = b2r[q];
      float4 v = make_float4(acc2[i][4 * q] * cs.x + bi.x, acc2[i][4 * q + 1] * cs.y + bi.y, acc2[i][4 * q + 2] * cs.z + bi.z, acc2[i][4 * q + 3] * cs.w + bi.w);
      v = relu4(v);
      uint2 hi, lo;
      encode4(v, hi, lo);
      ovf |= hi_nonfinite_bits(hi);
      unsigned char* d = smem + (i * 32 + l32) * PY + (ch >> 3) * 32 + h * 8;
      *reinterpret_cast<uint2*>(d) = hi; *reinterpret_cast<uint2*>(d + 16) = lo;
    }
  __syncthreads();

  // =========================== phase 3: out = relu(conv3(y2) + x); wave = channels [128 wave, +128) x both pixel tiles ===========================
  v16f acc3[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc3[i][j][r] = 0.f;
  v8h qxh[2][2], qxl[2][2];
#define SSG_B2_READ3(KT_, F_)                                                                                        \
  { _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                                  \
      const unsigned char* q_ = smem + (i * 32 + l32) * PY + ((KT_) * 2 + h) * 32;                                   \
      qxh[F_][i] = *reinterpret_cast<const v8h*>(q_); qxl[F_][i] = *reinterpret_cast<const v8h*>(q_ + 16); } }
  SSG_B2_READ3(0, 0)
#pragma unroll
  for (int kt = 0; kt < NK3; kt++) {
    const int slot = kt & 1;
    if (kt + 1 < NK3) {
      if (slot == 0) { SSG_B2_READ3(kt + 1, 1) } else { SSG_B2_READ3(kt + 1, 0) }
      if (kt >= 1) { if (slot == 0) { SSG_B2_LOADW3(kt + 1, 1) } else { SSG_B2_LOADW3(kt + 1, 0) } }      // (k-steps 0 and 1 were loaded before the y2 write)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r3h[slot][j], qxl[slot][i], acc3[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r3l[slot][j], qxh[slot][i], acc3[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r3h[slot][j], qxh[slot][i], acc3[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SSG_B2_READ3
#undef SSG_B2_LOADW3
  SSG_BN_STAMP(4)

  // ---- epilogue: per (pixel tile, channel tile) a 32 x 32 patch through this wave's LDS patch, then whole 128-byte row segments:
  // bias, residual (x, h8l8), ReLU, re-encode, store -- the code path of conv.hip / bottleneck.hip
  constexpr int CPR = 8, RPI = 8, ITS = 4;
  const int chunk = lane % CPR, prow = lane / CPR, odd = lane & 1;
  // (the patches lie behind y2, over y1's tail and the zero row, both dead since the barrier after phase 2: no barrier needed here)
  float* patch = reinterpret_cast<float*>(smem + PATCH_OFF) + wave * (32 * EP);
  const int64_t gpix = ((int64_t)img * p.H + ty0) * IW;     // first output pixel of this tile (its 64 pixels are contiguous)
  const float* __restrict__ resp = p.x + gpix * C + wave * 128 + chunk * 4;
  float* __restrict__ outp = p.out + gpix * C + wave * 128 + chunk * 4;
  float4 rr[2][ITS], b3r[2], cs3r[2];
  // operands of patch (i, j) = (e / 4, e % 4) are fetched while patch e - 1 is processed
#define SSG_B2_EPI_LOAD(E_, S_)                                                                                      \
  { const int64_t o_ = (int64_t)(((E_) / 4) * 32) * C + ((E_) % 4) * 32;                                             \
    _Pragma("unroll") for (int it = 0; it < ITS; it++) rr[S_][it] = *reinterpret_cast<const float4*>(resp + o_ + (int64_t)(it * RPI + prow) * C); \
    b3r[S_] = *reinterpret_cast<const float4*>(p.b3 + wave * 128 + ((E_) % 4) * 32 + chunk * 4);                     \
    cs3r[S_] = *reinterpret_cast<const float4*>(p.cs3 + wave * 128 + ((E_) % 4) * 32 + chunk * 4); }
  SSG_B2_EPI_LOAD(0, 0)
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int i = e / 4, j = e % 4, sl = e & 1;
    if (e + 1 < 8) SSG_B2_EPI_LOAD(e + 1, (e + 1) & 1)
    const float4 bias = b3r[sl], cs = cs3r[sl];
#pragma unroll
    for (int q = 0; q < 4; q++)
      *reinterpret_cast<float4*>(patch + l32 * EP + 8 * q + 4 * h) = make_float4(acc3[i][j][4 * q], acc3[i][j][4 * q + 1], acc3[i][j][4 * q + 2], acc3[i][j][4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int it = 0; it < ITS; it++) {
      const int pr = it * RPI + prow;
      float4 v = *reinterpret_cast<const float4*>(patch + pr * EP + chunk * 4);
      v.x = v.x * cs.x + bias.x; v.y = v.y * cs.y + bias.y; v.z = v.z * cs.z + bias.z; v.w = v.w * cs.w + bias.w;
      const float4 rw = rr[sl][it];
      float4 r4;
      {   // even lane holds hi0..7 of the 8-channel group, odd lane lo0..7; each needs hi and lo of ITS four channels
        const unsigned a0 = __float_as_uint(rw.x), a1 = __float_as_uint(rw.y), a2 = __float_as_uint(rw.z), a3 = __float_as_uint(rw.w);
        const unsigned g0 = (unsigned)__shfl_xor((int)(odd ? a0 : a2), 1, 64), g1 = (unsigned)__shfl_xor((int)(odd ? a1 : a3), 1, 64);
        const unsigned h0 = odd ? g0 : a0, h1 = odd ? g1 : a1, l0 = odd ? a2 : g0, l1 = odd ? a3 : g1;
        r4 = decode4(make_uint2(h0, h1), make_uint2(l0, l1));
      }
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      v = relu4(v);
      uint2 hp, lp;
      encode4(v, hp, lp);
      ovf |= hi_nonfinite_bits(hp);
      const unsigned rx = (unsigned)__shfl_xor((int)(odd ? hp.x : lp.x), 1, 64), ry = (unsigned)__shfl_xor((int)(odd ? hp.y : lp.y), 1, 64);
      const uint4 stv = make_uint4(odd ? rx : hp.x, odd ? ry : hp.y, odd ? lp.x : rx, odd ? lp.y : ry);
      *reinterpret_cast<uint4*>(outp + (int64_t)(i * 32 + pr) * C + j * 32) = stv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
#undef SSG_B2_EPI_LOAD
  if ((ovf & 0x80008000u) && p.overflow) *p.overflow = 1;
  SSG_BN_STAMP(5)
}

}  // namespace bneck2
}  // namespace ssg

// layer2 identity blocks (H x 16 x 512, mid 128) on 4-row tiles: H % 4 == 0 (bottleneck.hip's 8-wave kernel needs H % 8 == 0)
static int launch_bottleneck2(const ssg::bneck::Params& p, hipStream_t stream) {
  using namespace ssg::bneck2;
  static bool attr_set = false;
  if (!attr_set) {
    SSG_HIP(hipFuncSetAttribute((const void*)bottleneck2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(bottleneck2_kernel, dim3(p.B * (p.H / TH)), dim3(NTHR), LDS, stream, p);
  SSG_LAUNCH_CHECK("bottleneck2_kernel");
  return SSG_OK;
}
