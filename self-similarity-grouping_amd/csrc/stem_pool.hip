// stem_pool.hip -- the ResNet stem in ONE kernel: NCHW float32 images -> conv 7x7 / stride 2 (+ folded BatchNorm + ReLU)
// -> MaxPool 3x3 / stride 2 -> [B, H/4, W/4, 64] in the split-half (h8l8) format.
//
// reid/models/base.py:101-105 (conv1, bn1, relu, maxpool of the torchvision ResNet) behind reid/models/resnet.py:87-92, with
// the horizontal flip of reid/evaluators.py:12-16 folded into the image read.  Replaces three launches of conv.hip
// (nchw_to_nhwc4_h4l4, the stem instantiation of conv_igemm_kernel, maxpool3x3s2_h8l8): those move the 64-channel stem map
// (1 GB per 512 images) to HBM and back and gather every one of the 49 filter taps of every output pixel from global memory.
//
// One workgroup walks down a strip of pooled output rows (W/4 = 32 pixels each) of one image; per pooled row p:
//   * it needs 3 conv rows (2p-1 .. 2p+1), i.e. 11 image rows: the image patch is encoded to split halves ([4 x half hi]
//     [4 x half lo] per RGB0 pixel, as nchw_to_nhwc4_h4l4) and kept in a ring of 12 row slots in LDS (26 KB); a step down the
//     strip adds 4 rows, fetched into registers during the previous step's multiply;
//   * the implicit GEMM  conv[192 pixels, 64] = patch-taps[192, 224] x W^T  takes its pixel fragments from the LDS patch (a
//     filter tap is an address offset) and its weight fragments from registers (loaded once per wave): no barrier in the loop;
//   * the conv rows go to LDS as fp32 (a ring of three: a step computes rows 2p and 2p+1, row 2p-1 is the previous step's), the
//     3x3/2 max is taken there, the result is stored as h8l8.
// Numerics are those of the three launches bit for bit: the same k-steps (4 taps each, taps in r*7+s order, 49..55 zero),
// the same product order, bias / ReLU / encode, and the pool sees decode(encode(v)) like the separate kernel did.
#include "ssg_common.h"
#include <cstdlib>

namespace ssg {
namespace stem {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

constexpr int IW = 128, OW = 64, PW = 32;            // image / conv / pooled width
constexpr int PROWS = 11, PCOLS = 136;               // image patch: 11 rows, columns ix = pc - 4 (zero outside the image)
constexpr int RING = 12, ROWB = PCOLS * 16;          // the patch lives in a ring of 12 row slots (a step down the image adds 4 rows)
constexpr int CPITCH = 272;                          // conv tile: 192 pixels x 64 fp32 channels (+16 B)
constexpr int CONV_BYTES = 3 * OW * CPITCH;
constexpr int KSTEPS = 14, KROW = 224;               // 56 tap slots of 4 fp32-sized containers per weight row

__device__ __forceinline__ unsigned pk2(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ void enc4(const float4 v, uint2& hi, uint2& lo) {
  const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
  hi = make_uint2(pk2(h0, h1), pk2(h2, h3));
  lo = make_uint2(pk2((_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1)), pk2((_Float16)(v.z - (float)h2), (_Float16)(v.w - (float)h3)));
}
__device__ __forceinline__ float flo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float fhi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
__device__ __forceinline__ float4 dec4(const uint2 hi, const uint2 lo) {
  return make_float4(flo(hi.x) + flo(lo.x), fhi(hi.x) + fhi(lo.x), flo(hi.y) + flo(lo.y), fhi(hi.y) + fhi(lo.y));
}

#ifdef SSG_STEM_PROF
__device__ unsigned long long g_stem_prof[8];           // ticks per phase, summed over workgroups (thread 0)
#define SSG_STEM_T0() unsigned long long pt_ = __builtin_readcyclecounter();
#define SSG_STEM_ACC(I_) { const unsigned long long n_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_stem_prof[I_], n_ - pt_); pt_ = n_; }
#else
#define SSG_STEM_T0()
#define SSG_STEM_ACC(I_)
#endif

// one item of the patch = 4 pixels of one image row: three float4 (one per colour plane); rows / columns outside the image are
// zero padding (the loads then read a valid dummy address and the store writes zeros)
struct Item { float4 r, g, b; bool inside; };
__device__ __forceinline__ Item load_item(const float* ib, int64_t plane, int H, int iy, int g, int flip) {
  Item it;
  it.inside = iy >= 0 && iy < H && g >= 1 && g <= IW / 4;
  const int x0 = flip ? (IW - 4 * g) : (4 * g - 4);      // flipped image F[ix] = I[127 - ix]: the float4 at 128 - 4g, reversed
  const float* q = it.inside ? ib + (int64_t)iy * IW + x0 : ib;
  it.r = *reinterpret_cast<const float4*>(q); it.g = *reinterpret_cast<const float4*>(q + plane); it.b = *reinterpret_cast<const float4*>(q + 2 * plane);
  return it;
}
__device__ __forceinline__ void store_item(unsigned char* dst, const Item& it, int flip) {
  const float rr[4] = {it.r.x, it.r.y, it.r.z, it.r.w}, gg[4] = {it.g.x, it.g.y, it.g.z, it.g.w}, bb[4] = {it.b.x, it.b.y, it.b.z, it.b.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int s = flip ? 3 - k : k;
    uint2 hi, lo;
    enc4(make_float4(rr[s], gg[s], bb[s], 0.f), hi, lo);
    *reinterpret_cast<uint4*>(dst + k * 16) = it.inside ? make_uint4(hi.x, hi.y, lo.x, lo.y) : make_uint4(0u, 0u, 0u, 0u);
  }
}

// NT pixel tiles (32 pixels each, LDS byte offset acol[i] of the tile's first tap column) of one conv row whose 7 tap rows sit
// at roff[0..6] in the patch ring: acc[i] += taps x W^T over the 14 k-steps
template <int NT>
__device__ __forceinline__ void conv_row_gemm(const unsigned char* patch, const int (&acol)[NT], const int (&roff)[7], const v8h (&wh)[KSTEPS],
                                              const v8h (&wl)[KSTEPS], const int h, v16f (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < KSTEPS; t++) {
    // taps of this half-wave: 4t + 2h and the next one (slots 49..55 carry zero weights: any valid address will do)
    constexpr int NTAP = 49;
    const int ta0 = 4 * t < NTAP ? 4 * t : 0, tb0 = 4 * t + 1 < NTAP ? 4 * t + 1 : 0, ta1 = 4 * t + 2 < NTAP ? 4 * t + 2 : 0, tb1 = 4 * t + 3 < NTAP ? 4 * t + 3 : 0;
    const int oa = h ? roff[ta1 / 7] + (ta1 % 7) * 16 : roff[ta0 / 7] + (ta0 % 7) * 16;
    const int ob = h ? roff[tb1 / 7] + (tb1 % 7) * 16 : roff[tb0 / 7] + (tb0 % 7) * 16;
    v8h xh[NT], xl[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const v4f t0 = *reinterpret_cast<const v4f*>(patch + acol[i] + oa), t1 = *reinterpret_cast<const v4f*>(patch + acol[i] + ob);
      const v4f hi = {t0.x, t0.y, t1.x, t1.y}, lo = {t0.z, t0.w, t1.z, t1.w};
      xh[i] = __builtin_bit_cast(v8h, hi); xl[i] = __builtin_bit_cast(v8h, lo);
    }
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xl[i], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], xh[i], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xh[i], acc[i], 0, 0, 0);
  }
}

// one 32-pixel x 32-channel accumulator tile -> conv tile in LDS (fp32): bias, ReLU, and the encode / decode round trip the
// separate maxpool kernel saw; returns the non-finite bits of the encoded values
__device__ __forceinline__ unsigned tile_to_lds(const v16f& acc, const float* csb, unsigned char* dst_pixel, const int chbase) {
  unsigned ovf = 0u;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ch = chbase + 8 * q;
    const float4 c4 = *reinterpret_cast<const float4*>(csb + ch), b4 = *reinterpret_cast<const float4*>(csb + 64 + ch);
    float4 v = make_float4(acc[4 * q] * c4.x + b4.x, acc[4 * q + 1] * c4.y + b4.y, acc[4 * q + 2] * c4.z + b4.z, acc[4 * q + 3] * c4.w + b4.w);
    v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
    uint2 hi, lo;
    enc4(v, hi, lo);
    ovf |= ((hi.x & 0x7c007c00u) + 0x04000400u) | ((hi.y & 0x7c007c00u) + 0x04000400u);
    *reinterpret_cast<float4*>(dst_pixel + ch * 4) = dec4(hi, lo);
  }
  return ovf;
}

// A workgroup walks down a strip of `rs` pooled rows of one image: the weight fragments are loaded once, the image patch is
// a ring of 12 row slots in LDS (each step brings 4 new image rows, prefetched into registers under the previous step's
// multiply), the conv rows live in a ring of 3 row slots (a step computes the two new rows 2p, 2p+1; row 2p-1 is the
// previous step's).
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const float* __restrict__ img, int flip, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ cs, float* __restrict__ out, int B, int H, int rs, int* overflow) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[RING * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char ctile[CONV_BYTES];
  __shared__ __attribute__((aligned(16))) float csb[128];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
  const int OHP = H / 4, strips = (OHP + rs - 1) / rs;
  const int b = (int)blockIdx.x / strips, pr0 = ((int)blockIdx.x - b * strips) * rs, pr1 = min(pr0 + rs, OHP);
  SSG_STEM_T0()

  // ---- this wave: 32 output channels (j1) of ONE of the two new conv rows (nr) of a step, both 32-pixel halves of the row
  const int j1 = wave & 1, nr = __builtin_amdgcn_readfirstlane(wave >> 1);
  v8h wh[KSTEPS], wl[KSTEPS];                         // weight fragments, all 14 k-steps (half-wave h takes taps 4t+2h, 4t+2h+1)
  {
    const float* wr = w + (int64_t)(j1 * 32 + l32) * KROW + h * 8;
#pragma unroll
    for (int t = 0; t < KSTEPS; t++) {
      const v4f t0 = *reinterpret_cast<const v4f*>(wr + t * 16), t1 = *reinterpret_cast<const v4f*>(wr + t * 16 + 4);
      const v4f hi = {t0.x, t0.y, t1.x, t1.y}, lo = {t0.z, t0.w, t1.z, t1.w};
      wh[t] = __builtin_bit_cast(v8h, hi); wl[t] = __builtin_bit_cast(v8h, lo);
    }
  }
  if (tid < 128) csb[tid] = tid < 64 ? cs[tid] : bias[tid - 64];     // folded BatchNorm scale / bias: read from LDS after every multiply

  // ---- first window: image rows iy = 4*pr0 - 5 + py (py = 0..10) -> ring slot (rot + py) % 12, columns ix = pc - 4
  const int64_t plane = (int64_t)H * IW;
  const float* ib = img + (int64_t)b * 3 * plane;
  int rot = (4 * pr0 + 3) % RING;
  for (int it = tid; it < PROWS * (PCOLS / 4); it += 256) {
    const int py = it / (PCOLS / 4), g = it - py * (PCOLS / 4);
    const Item im = load_item(ib, plane, H, 4 * pr0 - 5 + py, g, flip);
    store_item(patch + ((rot + py) % RING) * ROWB + g * 64, im, flip);
  }
  __syncthreads();
  unsigned ovf = 0u;
  const int acol[2] = {(2 * l32 + 1) * 16, (2 * (32 + l32) + 1) * 16};   // tap (r, s) of a conv row = its r-th tap row slot, + s*16
  if (pr0 > 0) {
    // conv row 2*pr0 - 1 (window rows py = 0..6) is the previous strip's: recomputed here, one tile per wave
    int roff[7];
#pragma unroll
    for (int r = 0; r < 7; r++) roff[r] = ((rot + r) % RING) * ROWB;
    v16f a1[1];
#pragma unroll
    for (int r = 0; r < 16; r++) a1[0][r] = 0.f;
    const int ac1[1] = {nr ? acol[1] : acol[0]};      // (this wave's `nr` picks the half of the row here)
    conv_row_gemm<1>(patch, ac1, roff, wh, wl, h, a1);
    ovf |= tile_to_lds(a1[0], csb, ctile + (((2 * pr0 - 1) % 3) * OW + nr * 32 + l32) * CPITCH, j1 * 32 + 4 * h);
  }
  SSG_STEM_ACC(0)

  const int npy = tid / (PCOLS / 4), ng = tid - npy * (PCOLS / 4);      // next-window item of this thread (threads 0..135)
  const int pp = tid >> 3, cg = tid & 7;              // pool: (pooled pixel, 8 channels)
  for (int pr = pr0; pr < pr1; pr++) {
    const bool has_next = pr + 1 < pr1 && tid < 4 * (PCOLS / 4);
    // the 4 image rows the next window adds (its py = 7..10: iy = 4*pr + 6 + npy), in flight during the multiply
    const Item nx = load_item(ib, plane, H, has_next ? 4 * pr + 6 + npy : -1, ng, flip);
    int roff[7];                                     // conv row 2*pr + nr: window rows py = 2 + 2*nr + r
#pragma unroll
    for (int r = 0; r < 7; r++) roff[r] = ((rot + 2 + 2 * nr + r) % RING) * ROWB;
    v16f acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    conv_row_gemm<2>(patch, acol, roff, wh, wl, h, acc);
    __syncthreads();                                 // every wave is done with this window (and the previous step's pool with the conv rows)
    SSG_STEM_ACC(1)

    // next window's rows go to the slots this window no longer needs: py' = 7 + npy of rot' = rot + 4
    if (has_next) store_item(patch + ((rot + 11 + npy) % RING) * ROWB + ng * 64, nx, flip);
    {
      unsigned char* crow = ctile + (((2 * pr + nr) % 3) * OW) * CPITCH;
#pragma unroll
      for (int i = 0; i < 2; i++) ovf |= tile_to_lds(acc[i], csb, crow + (i * 32 + l32) * CPITCH, j1 * 32 + 4 * h);
    }
    __syncthreads();
    SSG_STEM_ACC(2)

    // MaxPool 3x3 / 2, pad 1 over conv rows 2*pr - 1 .. 2*pr + 1
    float4 m0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), m1 = m0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int c = 2 * pr - 1 + r;
      if (c < 0) continue;                           // (2*pr + 1 <= H/2 - 1 always)
      const unsigned char* crow = ctile + ((c % 3) * OW) * CPITCH;
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const int ox = 2 * pp - 1 + s;
        if (ox < 0) continue;
        const float* q = reinterpret_cast<const float*>(crow + ox * CPITCH) + cg * 8;
        const float4 a = *reinterpret_cast<const float4*>(q), c4 = *reinterpret_cast<const float4*>(q + 4);
        m0.x = fmaxf(m0.x, a.x); m0.y = fmaxf(m0.y, a.y); m0.z = fmaxf(m0.z, a.z); m0.w = fmaxf(m0.w, a.w);
        m1.x = fmaxf(m1.x, c4.x); m1.y = fmaxf(m1.y, c4.y); m1.z = fmaxf(m1.z, c4.z); m1.w = fmaxf(m1.w, c4.w);
      }
    }
    uint2 ha, la, hb, lb;
    enc4(m0, ha, la); enc4(m1, hb, lb);
    float* o = out + ((((int64_t)b * OHP + pr) * PW + pp) * 64 + cg * 8);
#ifdef SSG_STEM_NT_STORE        // A/B knob: the pooled map with the nt cache policy
    { typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
      const v4u_ s0_ = {ha.x, ha.y, hb.x, hb.y}, s1_ = {la.x, la.y, lb.x, lb.y};
      __builtin_nontemporal_store(s0_, reinterpret_cast<v4u_*>(o)); __builtin_nontemporal_store(s1_, reinterpret_cast<v4u_*>(o + 4)); }
#else
    *reinterpret_cast<uint4*>(o) = make_uint4(ha.x, ha.y, hb.x, hb.y);
    *reinterpret_cast<uint4*>(o + 4) = make_uint4(la.x, la.y, lb.x, lb.y);
#endif
    rot = (rot + 4) % RING;
    SSG_STEM_ACC(3)
  }
  if ((ovf & 0x80008000u) && overflow) *overflow = 1;
}

}  // namespace stem
}  // namespace ssg

// 1 when ssg_stem_pool_nchw_x has a kernel for this image shape
extern "C" int ssg_stem_pool_supported(int H, int W) { return (W == 128 && H >= 8 && H % 4 == 0) ? 1 : 0; }

// images [B,3,H,W] float32 NCHW (optionally mirrored left-right) -> relu(bn1(conv1)) -> maxpool, out [B,H/4,W/4,64] h8l8.
// w [64][224]: the stem weights as ssg_conv2d_nhwc_x takes them (Cin = 4 layout: per filter tap [4 x half hi][4 x half lo],
// taps in r*7+s order, zero-padded to 56 taps; rows pre-multiplied by powers of two that ch_scale undoes); bias fp32.
extern "C" int ssg_stem_pool_nchw_x(const float* images, int flip, const void* w, const float* bias, const float* ch_scale, void* out, int B, int H, int W,
                                    int32_t* overflow, hipStream_t stream) {
  if (B <= 0 || !ssg_stem_pool_supported(H, W) || !ch_scale) {
    ssg_set_error("ssg_stem_pool_nchw_x: unsupported shape B=%d H=%d W=%d (see ssg_stem_pool_supported)", B, H, W);
    return SSG_ERR_INVALID;
  }
  static int rs = -1;                          // pooled rows per workgroup (SSG_STEM_STRIP, tuning knob)
  if (rs < 0) { const char* e = getenv("SSG_STEM_STRIP"); rs = e ? atoi(e) : 16; if (rs < 1) rs = 1; }
  const int strips = (H / 4 + rs - 1) / rs;
  hipLaunchKernelGGL(ssg::stem::stem_pool_kernel, dim3(B * strips), dim3(256), 0, stream, images, flip, (const float*)w, bias, ch_scale, (float*)out, B, H,
                     rs, overflow);
  SSG_LAUNCH_CHECK("stem_pool_kernel");
  return SSG_OK;
}
