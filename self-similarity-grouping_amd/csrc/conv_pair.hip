// conv_pair.hip -- round 6, experimental (opt-in: SSG_CONV_PAIR=1): the tail of one identity bottleneck block and the head of the next as
// ONE launch,
//     out  = relu(conv3_1x1(y2) + bias3 + x)          (reid/models/base.py:84-90 of block b)
//     y1n  = relu(conv1_1x1(out) + bias1)             (reid/models/base.py:76-78 of block b + 1)
// for the layer3 shape (y2: 256 channels, out: 1024, y1n: 256).  The 1024-channel tensor `out` is still written (block b + 1 needs it as its
// residual) but not read back by the next block's first convolution: a workgroup owns 128 pixels, walks `out` in 128-channel chunks --
// GEMM1 (K = 256) into 32 accumulator registers, the epilogue of conv.hip (scale, bias, residual, ReLU, re-encode) whose 16-byte pieces go to
// HBM and, in the layout of an LDS-DMA stage, into a 64 KB stash -- and feeds each chunk straight into GEMM2 (K = 128 per chunk, 64
// accumulator registers that live across the chunks) with the pixel operand read from the stash.  One stream of k-tiles runs through
// four 16 KB stages (three tiles in flight) across both GEMMs and across the chunk boundary (weights and y2 do not depend on the epilogue), so no pipeline is
// refilled; every `vmcnt` literal below counts the vector-memory operations that are YOUNGER than the tile it publishes.
// Same three-product multiply, same k order and the same epilogue arithmetic as the two launches it replaces: bit-identical
// (tests/test_gpu_parity.py::test_conv_pair_matches_the_two_launches).  Whether it is FASTER is a measurement: DESIGN.md section 11.
#include "ssg_common.h"

namespace ssg {
namespace pairk {

struct PairParams {
  const float* y2; const float* w3; const float* b3; const float* cs3; const float* res; float* out;
  const float* w1n; const float* b1n; const float* cs1n; float* y1n;
  int M; int* overflow;
  unsigned long long* prof = nullptr;   // SSG_PAIR_PROF builds (tools/micro/pair_prof.hip): ticks per workgroup -- total, GEMM1, chunk epilogues, GEMM2, last epilogue
};
#ifdef SSG_PAIR_PROF
#define SSG_PAIR_STAMP(ACC_) { const unsigned long long t_ = __builtin_readcyclecounter(); ACC_ += t_ - tprev; tprev = t_; }
#else
#define SSG_PAIR_STAMP(ACC_)
#endif

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, unsigned lds_addr, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, voff, 0, 0, 0);
}

template <int K1, int C, int N2>
__global__ __launch_bounds__(512, 2) void conv_pair_kernel(PairParams p) {
  static_assert(K1 == 256 && C % 128 == 0 && N2 == 256, "layer3 shape: 16 k-tiles for GEMM1, 8 per chunk for GEMM2, 256 next-conv1 channels");
  constexpr int NCH = C / 128;                       // chunks of 128 `out` channels
  constexpr int EP = 32, RPI = 8, ITS = 4;           // epilogue patches unpadded (32 x 32 fp32 = 4 KB per wave), 16-byte slots XOR-swizzled with (row >> 1) & 7
  __shared__ __attribute__((aligned(1024))) unsigned char S[8 * 128 * 64];        // the chunk of `out` as GEMM2's pixel operand: [k-tile][row][64 B], swizzled like a stage
  __shared__ __attribute__((aligned(1024))) unsigned char st0[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char st1[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char st2[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char st3[16384];             // 64 + 4 x 16 + 32 KB = the CU's 160 KB: one workgroup per CU
  __shared__ __attribute__((aligned(16))) float patches[8 * 32 * EP];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;
  const int tm = (int)blockIdx.x, m0 = tm * 128;
  const unsigned out_bytes = (unsigned)((int64_t)p.M * C * 4);
  const __amdgpu_buffer_rsrc_t y2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.y2), 0, (unsigned)((int64_t)p.M * K1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, (unsigned)(C * K1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1n), 0, (unsigned)(N2 * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t y1_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y1n, 0, (unsigned)((int64_t)p.M * N2 * 4), 0x00020000);

  // ---- DMA addressing: one instruction = 16 rows x 64 B; lane (drow, pc) fetches logical chunk pc ^ g(drow) into physical slot pc
  const int drow = lane >> 2, pc = lane & 3, lc4 = (pc ^ ((drow >> 2) & 3)) * 4;
  const int arow = m0 + wave * 16 + drow;
  const unsigned a_off0 = arow < p.M ? (unsigned)((arow * K1 + lc4) * 4) : 0x80000000u;           // + 64 per k-tile
  const unsigned w3_off0 = (unsigned)(((wave * 16 + drow) * K1 + lc4) * 4);                      // + chunk * 128 * K1 * 4 + 64 per k-tile
  const unsigned w1_off0 = (unsigned)(((wave * 16 + drow) * C + lc4) * 4), w1_off1 = w1_off0 + (unsigned)(128 * C * 4);   // + chunk * 512 + 64 per k-tile
  const unsigned st_base[4] = {(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)st0, (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)st1,
                               (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)st2, (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)st3};
  // tile T of chunk CC (0..15: GEMM1 k-tile T; 16..23: GEMM2 k-tile T - 16) into stage T % 4: two instructions per wave either way
#define PAIR_DMA(T, CC)                                                                                              \
  { if constexpr ((T) < 16) {                                                                                        \
      dma16(y2_rsrc, st_base[(T) % 4] + (unsigned)wave * 1024u, a_off0 == 0x80000000u ? 0x80000000u : a_off0 + (unsigned)(T) * 64u);            \
      dma16(w3_rsrc, st_base[(T) % 4] + 8192u + (unsigned)wave * 1024u, w3_off0 + (unsigned)(CC) * (unsigned)(128 * K1 * 4) + (unsigned)(T) * 64u); \
    } else {                                                                                                         \
      dma16(w1_rsrc, st_base[(T) % 4] + (unsigned)wave * 1024u, w1_off0 + (unsigned)(CC) * 512u + (unsigned)((T) - 16) * 64u);                  \
      dma16(w1_rsrc, st_base[(T) % 4] + (unsigned)(wave + 8) * 1024u, w1_off1 + (unsigned)(CC) * 512u + (unsigned)((T) - 16) * 64u);            \
    } }

  // ---- fragment addressing (as conv_dma_kernel): lane = (row l32 of a 32-row MFMA tile, k half h), swizzle by (row >> 2) & 3
  const int g = (l32 >> 2) & 3;
  const int offh = ((2 * h) ^ g) * 16, offl = ((2 * h + 1) ^ g) * 16;
  const int arow_b = (wm * 64 + l32) * 64;                          // this lane's first pixel row in a stage / in the stash
  const int b1row_b = 8192 + (wn * 32 + l32) * 64;                  // GEMM1: W3 rows of this wave's 32 chunk channels
  const int b2row_b = (wn * 64 + l32) * 64;                         // GEMM2: W1 rows of this wave's 64 channels
  v16f acc1[2], acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc2[i][j][r] = 0.f;

  // GEMM1 tile in stage ST: 2 x 1 MFMA tiles, three products; GEMM2 tile: pixel fragments from the stash (k-tile KT), 2 x 2 MFMA tiles
#define PAIR_MMA1(ST)                                                                                                \
  { v8h ah_[2], al_[2], bh_, bl_;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                                  \
      ah_[i] = *reinterpret_cast<const v8h*>(ST + arow_b + i * 2048 + offh); al_[i] = *reinterpret_cast<const v8h*>(ST + arow_b + i * 2048 + offl); } \
    bh_ = *reinterpret_cast<const v8h*>(ST + b1row_b + offh); bl_ = *reinterpret_cast<const v8h*>(ST + b1row_b + offl);  \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, al_[i], acc1[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl_, ah_[i], acc1[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; i++) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_, ah_[i], acc1[i], 0, 0, 0); }
#define PAIR_MMA2(ST, KT)                                                                                            \
  { v8h ah_[2], al_[2], bh_[2], bl_[2];                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                                  \
      ah_[i] = *reinterpret_cast<const v8h*>(S + (KT) * 8192 + arow_b + i * 2048 + offh); al_[i] = *reinterpret_cast<const v8h*>(S + (KT) * 8192 + arow_b + i * 2048 + offl); } \
    _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                                  \
      bh_[j] = *reinterpret_cast<const v8h*>(ST + b2row_b + j * 2048 + offh); bl_[j] = *reinterpret_cast<const v8h*>(ST + b2row_b + j * 2048 + offl); } \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 2; j++)                      \
      acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_[j], al_[i], acc2[i][j], 0, 0, 0);                      \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 2; j++)                      \
      acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl_[j], ah_[i], acc2[i][j], 0, 0, 0);                      \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 2; j++)                      \
      acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh_[j], ah_[i], acc2[i][j], 0, 0, 0); }
#ifdef SSG_PAIR_ABL_NOMMA        // ablation build (tools/micro/pair_prof.hip): the DMA stream, barriers and epilogues without a multiply -- wrong results
#undef PAIR_MMA1
#undef PAIR_MMA2
#define PAIR_MMA1(ST)
#define PAIR_MMA2(ST, KT)
#endif
  // publish tile t: VM = vector-memory operations of this wave that are younger than the tile's DMA; lgkmcnt(0): "I am done reading the
  // stage the next DMA overwrites" (conv_dma_kernel's protocol)
#define PAIR_PUBLISH(VM) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
#define PAIR_STAGE(T) ((T) % 4 == 0 ? st0 : ((T) % 4 == 1 ? st1 : ((T) % 4 == 2 ? st2 : st3)))
#if defined(SSG_PAIR_ABL_NOEPI) || defined(SSG_PAIR_ABL_NORES)
#define PAIR_VML(N_, ABL_) ABL_    // ablation builds: without the epilogue's 10 loads in the queue
#else
#define PAIR_VML(N_, ABL_) N_
#endif
#if defined(SSG_PAIR_ABL_NOEPI) || defined(SSG_PAIR_ABL_NOSTORE)
#define PAIR_VMS(N_, ABL_) ABL_    // ... without its 8 stores
#else
#define PAIR_VMS(N_, ABL_) N_
#endif
#ifndef SSG_PAIR_AUX
#define SSG_PAIR_AUX 2             // cache-policy bits of the residual loads and the stores of `out`: nt -- streamed once, they need no line of the L2
#endif                             // the k-tile stream goes through (0.631 -> 0.470 ms at nearly equal fabric traffic: profiles/r06_conv_pair_phases.txt)
#ifndef SSG_PAIR_AUX_Y1
#define SSG_PAIR_AUX_Y1 0          // ... of the stores of y1n
#endif

  // ---- epilogue addressing (conv_dma_kernel's straight-line epilogue)
  float* patch = patches + wave * (32 * EP);
  const int chunk = lane & 7, prow = lane >> 3, odd = lane & 1;
  const int psw = (l32 >> 1) & 7;                     // patch swizzle of the row this lane writes (rows r: slot ^ ((r >> 1) & 7))
  const unsigned rowb = (unsigned)C * 4u;
  const unsigned lane_off = (unsigned)prow * rowb + (unsigned)chunk * 16u;
  const unsigned ubase1 = (unsigned)(m0 + wm * 64) * rowb + (unsigned)(wn * 32) * 4u;          // + chunk index * 512 + (i * 32 + it * 8) * rowb
  unsigned ovf = 0u;

#ifdef SSG_PAIR_PROF
  unsigned long long tprev = __builtin_readcyclecounter(), t_g1 = 0, t_ep = 0, t_g2 = 0, t_last = 0;
  const unsigned long long tstart = tprev;
#endif
  PAIR_DMA(0, 0)
  PAIR_DMA(1, 0)
  PAIR_DMA(2, 0)
  for (int c = 0; c < NCH; c++) {
    const int cn = c + 1 < NCH ? c + 1 : c;            // the chunk whose first tiles are requested at the end of this one (clamped: a harmless re-fetch)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[i][r] = 0.f;
    v4u rr[2][ITS];
    v4f bias1, cs1;
    // ======== GEMM1: tiles 0 .. 15.  Step t publishes tile t and requests tile t + 3 into the stage of tile t - 1 (three tiles in flight)
#define PAIR_STEP1(T, VM)                                                                                            \
    { PAIR_PUBLISH(VM)                                                                                               \
      PAIR_DMA((T) + 3, c)                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      PAIR_MMA1(PAIR_STAGE(T)) }
    PAIR_STEP1(0, 4) PAIR_STEP1(1, 4) PAIR_STEP1(2, 4) PAIR_STEP1(3, 4) PAIR_STEP1(4, 4) PAIR_STEP1(5, 4) PAIR_STEP1(6, 4) PAIR_STEP1(7, 4)
    PAIR_STEP1(8, 4) PAIR_STEP1(9, 4) PAIR_STEP1(10, 4) PAIR_STEP1(11, 4)
    {   // tile 12: behind its DMA, the epilogue's operands (8 residual pieces, bias, scale: 10 loads that stay in flight over tiles 12 .. 15)
      PAIR_PUBLISH(4)
      PAIR_DMA(15, c)
      {
        const int col = c * 128 + wn * 32 + chunk * 4;
        bias1 = *reinterpret_cast<const v4f*>(p.b3 + col); cs1 = *reinterpret_cast<const v4f*>(p.cs3 + col);
#ifndef SSG_PAIR_ABL_NOEPI       // ablation build: no chunk epilogue (no residual read, no store of `out`, stale stash) -- wrong results
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int it = 0; it < ITS; it++)
#ifdef SSG_PAIR_ABL_NORES
            rr[i][it] = v4u{0u, 0u, 0u, 0u};
#else
            rr[i][it] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, ubase1 + (unsigned)c * 512u + (unsigned)(i * 32 + it * RPI) * rowb + lane_off, 0, SSG_PAIR_AUX);
#endif
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      PAIR_MMA1(PAIR_STAGE(12))
    }
    PAIR_STEP1(13, PAIR_VML(14, 4))      // younger than tile 13: tiles 14, 15 (4) + the 10 loads
    PAIR_STEP1(14, PAIR_VML(14, 4))      // tile 15 (2) + 10 loads + tile 16 (2)
    PAIR_STEP1(15, PAIR_VML(14, 4))      // 10 loads + tiles 16, 17 (4)
#undef PAIR_STEP1
    // (no barrier here: the stage of tile 15 is next written by the DMA of tile 19, behind the barrier of step 16; the stash was last read
    // 16 barriers ago)
    SSG_PAIR_STAMP(t_g1)
    // ======== epilogue of conv3 for this chunk: out -> HBM and -> the stash
#ifndef SSG_PAIR_ABL_NOEPI
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4*>(patch + l32 * EP + (((2 * q + h) ^ psw) * 4)) = make_float4(acc1[i][4 * q], acc1[i][4 * q + 1], acc1[i][4 * q + 2], acc1[i][4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        float4 v = *reinterpret_cast<const float4*>(patch + (it * RPI + prow) * EP + ((chunk ^ ((it & 1) * 4 + (prow >> 1))) * 4));
        v.x = v.x * cs1[0] + bias1[0]; v.y = v.y * cs1[1] + bias1[1]; v.z = v.z * cs1[2] + bias1[2]; v.w = v.w * cs1[3] + bias1[3];
        {
          const unsigned a0 = rr[i][it][0], a1 = rr[i][it][1], a2 = rr[i][it][2], a3 = rr[i][it][3];
          const unsigned g0 = lane_xor1(odd ? a0 : a2), g1 = lane_xor1(odd ? a1 : a3);
          const float4 r4 = split_decode4(make_uint2(odd ? g0 : a0, odd ? g1 : a1), make_uint2(odd ? a2 : g0, odd ? a3 : g1));
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        uint2 hp, lp;
        split_encode4(v, hp, lp);
        ovf |= ((hp.x & 0x7c007c00u) + 0x04000400u) | ((hp.y & 0x7c007c00u) + 0x04000400u);
        const unsigned rx = lane_xor1(odd ? hp.x : lp.x), ry = lane_xor1(odd ? hp.y : lp.y);
        const v4u stv = {odd ? rx : hp.x, odd ? ry : hp.y, odd ? lp.x : rx, odd ? lp.y : ry};
#ifndef SSG_PAIR_ABL_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b128(stv, out_rsrc, ubase1 + (unsigned)c * 512u + (unsigned)(i * 32 + it * RPI) * rowb + lane_off, 0, SSG_PAIR_AUX);
#endif
        // the same 16 bytes as a piece of GEMM2's pixel operand: row R of the tile, k-tile wn * 2 + chunk / 4, logical 16-byte slot chunk % 4
        const int R = wm * 64 + i * 32 + it * RPI + prow;
        *reinterpret_cast<v4u*>(S + (wn * 2 + (chunk >> 2)) * 8192 + R * 64 + (((chunk & 3) ^ ((R >> 2) & 3)) * 16)) = stv;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#endif
    SSG_PAIR_STAMP(t_ep)
    // ======== GEMM2: tiles 16 .. 23 (k-tiles 0 .. 7 of this chunk); the first barrier also publishes the stash
#define PAIR_STEP2(T, VM, DMA_)                                                                                      \
    { PAIR_PUBLISH(VM)                                                                                               \
      DMA_                                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      PAIR_MMA2(PAIR_STAGE(T), (T) - 16) }
    PAIR_STEP2(16, PAIR_VMS(12, 4), PAIR_DMA(19, c))      // younger than tile 16: tiles 17, 18 (4) + the 8 stores
    PAIR_STEP2(17, PAIR_VMS(12, 4), PAIR_DMA(20, c))      // tile 18 (2) + 8 stores + tile 19 (2)
    PAIR_STEP2(18, PAIR_VMS(12, 4), PAIR_DMA(21, c))      // 8 stores + tiles 19, 20 (4)
    PAIR_STEP2(19, 4, PAIR_DMA(22, c))
    PAIR_STEP2(20, 4, PAIR_DMA(23, c))
    PAIR_STEP2(21, 4, PAIR_DMA(0, cn))
    PAIR_STEP2(22, 4, PAIR_DMA(1, cn))
    PAIR_STEP2(23, 4, PAIR_DMA(2, cn))
#undef PAIR_STEP2
    SSG_PAIR_STAMP(t_g2)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ======== epilogue of the next block's conv1: y1n = relu(acc2 * cs + bias), four 32 x 32 patches per wave
  {
    const unsigned rowb2 = (unsigned)N2 * 4u;
    const unsigned lane_off2 = (unsigned)prow * rowb2 + (unsigned)chunk * 16u;
    const unsigned ubase2 = (unsigned)(m0 + wm * 64) * rowb2 + (unsigned)(wn * 64) * 4u;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const v4f bias = *reinterpret_cast<const v4f*>(p.b1n + wn * 64 + j * 32 + chunk * 4), cs = *reinterpret_cast<const v4f*>(p.cs1n + wn * 64 + j * 32 + chunk * 4);
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(patch + l32 * EP + (((2 * q + h) ^ psw) * 4)) = make_float4(acc2[i][j][4 * q], acc2[i][j][4 * q + 1], acc2[i][j][4 * q + 2], acc2[i][j][4 * q + 3]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int it = 0; it < ITS; it++) {
          float4 v = *reinterpret_cast<const float4*>(patch + (it * RPI + prow) * EP + ((chunk ^ ((it & 1) * 4 + (prow >> 1))) * 4));
          v.x = v.x * cs[0] + bias[0]; v.y = v.y * cs[1] + bias[1]; v.z = v.z * cs[2] + bias[2]; v.w = v.w * cs[3] + bias[3];
          v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
          uint2 hp, lp;
          split_encode4(v, hp, lp);
          ovf |= ((hp.x & 0x7c007c00u) + 0x04000400u) | ((hp.y & 0x7c007c00u) + 0x04000400u);
          const unsigned rx = lane_xor1(odd ? hp.x : lp.x), ry = lane_xor1(odd ? hp.y : lp.y);
          const v4u stv = {odd ? rx : hp.x, odd ? ry : hp.y, odd ? lp.x : rx, odd ? lp.y : ry};
          __builtin_amdgcn_raw_buffer_store_b128(stv, y1_rsrc, ubase2 + (unsigned)(i * 32 + it * RPI) * rowb2 + (unsigned)j * 128u + lane_off2, 0, SSG_PAIR_AUX_Y1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
  if ((ovf & 0x80008000u) && p.overflow) *p.overflow = 1;
#ifdef SSG_PAIR_PROF
  SSG_PAIR_STAMP(t_last)
  if (p.prof && tid == 0) {
    unsigned long long* q = p.prof + (size_t)blockIdx.x * 8;
    q[0] = tprev - tstart; q[1] = t_g1; q[2] = t_ep; q[3] = t_g2; q[4] = t_last;
  }
#endif
#undef PAIR_DMA
#undef PAIR_MMA1
#undef PAIR_MMA2
#undef PAIR_PUBLISH
#undef PAIR_STAGE
#undef PAIR_VML
#undef PAIR_VMS
}

}  // namespace pairk
}  // namespace ssg

// 1 when ssg_conv_pair_nhwc_x has a kernel for this pair of 1x1 convolutions (K1 -> C with residual, then C -> N2)
extern "C" int ssg_conv_pair_supported(int K1, int C, int N2) { return K1 == 256 && C == 1024 && N2 == 256 ? 1 : 0; }

// out [M, C] = relu(y2 [M, K1] * w3^T * cs3 + b3 + res [M, C]);  y1n [M, N2] = relu(out * w1n^T * cs1n + b1n); every tensor h8l8 (split-half),
// weights as ssg_conv2d_nhwc_x takes them (w3 [C][K1], w1n [N2][C], rows pre-multiplied by the powers of two cs* undo).  out must not alias res.
extern "C" int ssg_conv_pair_nhwc_x(const void* y2, const void* w3, const float* b3, const float* cs3, const void* res, void* out,
                                    const void* w1n, const float* b1n, const float* cs1n, void* y1n, int M, int K1, int C, int N2,
                                    int32_t* overflow, hipStream_t stream) {
  using namespace ssg::pairk;
  if (M <= 0 || !ssg_conv_pair_supported(K1, C, N2) || !cs3 || !cs1n || !res || out == res || ((int64_t)M + 128) * C * 4 >= (int64_t)0xffffffff) {
    ssg_set_error("ssg_conv_pair_nhwc_x: unsupported pair M=%d K1=%d C=%d N2=%d (see ssg_conv_pair_supported)", M, K1, C, N2);
    return SSG_ERR_INVALID;
  }
  PairParams p;
  p.y2 = (const float*)y2; p.w3 = (const float*)w3; p.b3 = b3; p.cs3 = cs3; p.res = (const float*)res; p.out = (float*)out;
  p.w1n = (const float*)w1n; p.b1n = b1n; p.cs1n = cs1n; p.y1n = (float*)y1n; p.M = M; p.overflow = overflow;
  hipLaunchKernelGGL((conv_pair_kernel<256, 1024, 256>), dim3((M + 127) / 128), dim3(512), 0, stream, p);
  SSG_LAUNCH_CHECK("conv_pair_kernel");
  return SSG_OK;
}
