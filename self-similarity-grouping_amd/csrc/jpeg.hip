// jpeg.hip -- baseline JPEG decode on the GPU, bit-exact with what `Image.open(fpath).convert('RGB')` returns
// (reid/utils/data/preprocessor.py:28: Pillow -> libjpeg(-turbo) with its default decompression parameters).
//
// The reference decodes every image of the extraction loaders on the CPU inside DataLoader workers.  Here a batch of files is
// parsed on the host (marker segments only: ssg_amd/jpeg.py) and decoded in three launches:
//   huffman_kernel   one thread per restart segment (one per image when the file has no DRI marker): jdhuff.c's sequential
//                    decoder -- DC prediction, run/size symbols through an 8-bit look-ahead table with the bit-serial
//                    maxcode walk behind it, byte-stuffing removed on the fly -- writing the non-zero quantised coefficients
//                    in natural order.  Sequential per segment by nature; the parallelism is across the images of a batch;
//   idct_kernel      one thread per 8x8 block: dequantisation + the "islow" integer inverse DCT of jidctint.c
//                    (CONST_BITS = 13, PASS1_BITS = 2, the library's default) + its range-limit table;
//   colour_kernel    one thread per pixel: "fancy" (triangle filter) chroma upsampling of jdsample.c with the context rows of
//                    jdmainct.c (h2v1 / h2v2; plain replication for components at most 2 samples wide) and the 16-bit
//                    fixed-point YCbCr -> RGB conversion of jdcolor.c; grayscale is replicated like convert('RGB') does.
// Everything is integer arithmetic restated from the published algorithms, so the output bytes equal Pillow's
// (tests: oracle/jpeg_oracle.py pinned against Pillow on the CPU, this file against both on the GPU).
// Files outside the supported class -- progressive / arithmetic / 12 bit / CMYK / unusual sampling factors -- are decoded by
// Pillow on the host as before (ssg_amd/jpeg.py reports how many).
#include "ssg_common.h"

namespace ssg {
namespace jpeg {

constexpr int IMG_WORDS = 8 + 3 * 8;      // int64 words per image descriptor (include/ssg_hip.h, ssg_jpeg_decode_batch)
constexpr int SEG_WORDS = 5;
enum { I_W = 0, I_H, I_NCOMP, I_HS, I_VS, I_MCUX, I_MCUY, I_OUT, I_COMP0 };
enum { C_COEF = 0, C_BW, C_BH, C_PLANE, C_PITCH, C_QT, C_DC, C_AC };

__constant__ unsigned char c_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Tables {
  const uint16_t* look;      // [ntab][256]: (length << 8) | symbol for codes of at most 8 bits, 0 otherwise
  const int32_t* maxcode;    // [ntab][18]: largest code of each length (-1: none), [17] = 0xFFFFF
  const int32_t* valoff;     // [ntab][17]: vals index of the first code of each length minus that code
  const uint8_t* vals;       // [ntab][256]
};

// bit reader over one restart segment: bytes are fetched 8 at a time with the next chunk already on its way; a 0xFF00 pair
// is the stuffed byte 0xFF, any other 0xFF.. pair is a marker (the segment ends: zeros are fed, like libjpeg does)
struct BitReader {
  const uint8_t* base; int64_t pos, end;
  uint64_t cur, nxt; int64_t chunk;       // cur = bytes [8*chunk, 8*chunk+8) of the pool, nxt the following eight
  uint64_t acc; int n;
  int pad;                                // zero bits fed since the data ran out (or a marker was hit): consuming any of them = a short segment
  __device__ __forceinline__ void init(const uint8_t* pool, int64_t off, int64_t len) {
    base = pool; pos = off; end = off + len; chunk = off >> 3;
    cur = *reinterpret_cast<const uint64_t*>(base + (chunk << 3)); nxt = *reinterpret_cast<const uint64_t*>(base + (chunk << 3) + 8);
    acc = 0; n = 0; pad = 0;
  }
  __device__ __forceinline__ unsigned byte_at(int64_t q) {     // q in the current or the next chunk
    const int64_t c = q >> 3;
    if (c != chunk) { cur = nxt; chunk = c; nxt = *reinterpret_cast<const uint64_t*>(base + (chunk << 3) + 8); }
    return (unsigned)(cur >> ((q & 7) * 8)) & 0xffu;
  }
  __device__ __forceinline__ void fill() {
    while (n <= 48) {
      unsigned b = 0;
      if (pos < end) {
        b = byte_at(pos); pos++;
        if (b == 0xffu) {
          const unsigned nx = pos < end ? byte_at(pos) : 1u;
          if (nx == 0) pos++;
          else { pos = end; b = 0; pad += 8; }
        }
      } else pad += 8;
      acc = (acc << 8) | b; n += 8;
    }
  }
  __device__ __forceinline__ unsigned peek(int k) const { return (unsigned)(acc >> (n - k)) & ((1u << k) - 1u); }
  __device__ __forceinline__ unsigned get(int k) { const unsigned v = peek(k); n -= k; return v; }
};

__device__ __forceinline__ int decode_symbol(BitReader& br, const Tables& t, int tab) {
  const unsigned lk = t.look[tab * 256 + br.peek(8)];
  if (lk) { br.n -= (int)(lk >> 8); return (int)(lk & 0xffu); }
  const int32_t* mc = t.maxcode + tab * 18;
  int l = 9;
  int code = (int)br.peek(9);
  while (l <= 16 && code > mc[l]) { l++; code = (int)br.peek(l); }
  if (l > 16) { br.n -= 16; return 0; }                      // a code no table holds (corrupt data): libjpeg returns 0 as well
  br.n -= l;
  return (int)t.vals[tab * 256 + ((code + t.valoff[tab * 17 + l]) & 0xff)];
}
__device__ __forceinline__ int extend(int r, int s) { return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r; }

__global__ __launch_bounds__(64) void huffman_kernel(const uint8_t* __restrict__ ecs, const int64_t* __restrict__ segs, int nseg,
                                                     const int64_t* __restrict__ imgs, Tables t, int16_t* __restrict__ coef,
                                                     int32_t* __restrict__ status) {
  const int si = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (si >= nseg) return;
  const int64_t* sd = segs + (int64_t)si * SEG_WORDS;
  const int64_t* im = imgs + sd[0] * IMG_WORDS;
  const int ncomp = (int)im[I_NCOMP], mcux = (int)im[I_MCUX];
  BitReader br;
  br.init(ecs, sd[1], sd[2]);
  int pred[3] = {0, 0, 0};
  int bad = 0;                            // 1: a run pushed the coefficient index past 63 (libjpeg warns and writes elsewhere), 2: segment ran out of data
  const int mcu0 = (int)sd[3], mcu1 = mcu0 + (int)sd[4];
  for (int mcu = mcu0; mcu < mcu1; mcu++) {
    const int my = mcu / mcux, mx = mcu - my * mcux;
    for (int ci = 0; ci < ncomp; ci++) {
      const int64_t* cd = im + I_COMP0 + ci * 8;
      const int hs = ci == 0 ? (int)im[I_HS] : 1, vs = ci == 0 ? (int)im[I_VS] : 1;
      const int bw = (int)cd[C_BW], dct = (int)cd[C_DC], act = (int)cd[C_AC];
      for (int by = 0; by < vs; by++)
        for (int bx = 0; bx < hs; bx++) {
          int16_t* blk = coef + (cd[C_COEF] + (int64_t)(my * vs + by) * bw + (mx * hs + bx)) * 64;
          br.fill();
          const int s = decode_symbol(br, t, dct);
          if (s) pred[ci] += extend((int)br.get(s), s);
          blk[0] = (int16_t)pred[ci];
          for (int k = 1; k < 64;) {
            br.fill();
            const int rs = decode_symbol(br, t, act), r = rs >> 4, sz = rs & 15;
            if (sz) {
              k += r;
              bad |= (k > 63);
              blk[c_zigzag[k & 63]] = (int16_t)extend((int)br.get(sz), sz);
              k++;
            } else if (r == 15) k += 16;
            else break;
          }
        }
    }
  }
  // the host validated the tables (DC categories <= 15, jpeg.py), so every shift above is defined; what cannot be known before decoding
  // is flagged per image and the host hands those files to Pillow (whose behaviour on damaged data is the reference's)
  if (br.n < br.pad) bad |= 2;
  if (bad) atomicOr(status + sd[0], bad);
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// jdmaster.c prepare_range_limit_table, post-IDCT half, index x & 1023
__device__ __forceinline__ unsigned range_limit_idct(int x) {
  const int i = x & 1023;
  return (unsigned)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}
// one 8-point pass of jpeg_idct_islow (even part, odd part), results descaled by `sh`
__device__ __forceinline__ void idct8(const int v0, const int v1, const int v2, const int v3, const int v4, const int v5, const int v6, const int v7,
                                      const int sh, int (&o)[8]) {
  int z1 = (v2 + v6) * 4433;
  const int tmp2 = z1 + v6 * (-15137), tmp3 = z1 + v2 * 6270;
  const int tmp0 = (v0 + v4) << 13, tmp1 = (v0 - v4) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int t0 = v7, t1 = v5, t2 = v3, t3 = v1;
  z1 = t0 + t3;
  int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
  const int z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  o[0] = descale(tmp10 + t3, sh); o[7] = descale(tmp10 - t3, sh);
  o[1] = descale(tmp11 + t2, sh); o[6] = descale(tmp11 - t2, sh);
  o[2] = descale(tmp12 + t1, sh); o[5] = descale(tmp12 - t1, sh);
  o[3] = descale(tmp13 + t0, sh); o[4] = descale(tmp13 - t0, sh);
}

// grid.y = image * 3 + component, grid.x * 64 threads over the component's blocks
__global__ __launch_bounds__(64) void idct_kernel(const int64_t* __restrict__ imgs, const int16_t* __restrict__ coef, const uint16_t* __restrict__ qts,
                                                  uint8_t* __restrict__ planes) {
  const int img = (int)blockIdx.y / 3, ci = (int)blockIdx.y % 3;
  const int64_t* im = imgs + (int64_t)img * IMG_WORDS;
  if (ci >= (int)im[I_NCOMP]) return;
  const int64_t* cd = im + I_COMP0 + ci * 8;
  const int bw = (int)cd[C_BW], bh = (int)cd[C_BH];
  const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (b >= bw * bh) return;
  const int16_t* blk = coef + (cd[C_COEF] + b) * 64;
  const uint16_t* qt = qts + cd[C_QT] * 64;
  int ws[8][8];
  // pass 1: columns (index = row), dequantised input
#pragma unroll
  for (int c = 0; c < 8; c++) {
    int v[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = (int)blk[r * 8 + c] * (int)qt[r * 8 + c];
    idct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], 13 - 2, o);
#pragma unroll
    for (int r = 0; r < 8; r++) ws[r][c] = o[r];
  }
  const int by = b / bw, bx = b - by * bw;
  uint8_t* dst = planes + cd[C_PLANE] + (int64_t)(by * 8) * cd[C_PITCH] + bx * 8;
  // pass 2: rows
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int o[8];
    idct8(ws[r][0], ws[r][1], ws[r][2], ws[r][3], ws[r][4], ws[r][5], ws[r][6], ws[r][7], 13 + 2 + 3, o);
    const unsigned lo = range_limit_idct(o[0]) | (range_limit_idct(o[1]) << 8) | (range_limit_idct(o[2]) << 16) | (range_limit_idct(o[3]) << 24);
    const unsigned hi = range_limit_idct(o[4]) | (range_limit_idct(o[5]) << 8) | (range_limit_idct(o[6]) << 16) | (range_limit_idct(o[7]) << 24);
    *reinterpret_cast<uint2*>(dst + (int64_t)r * cd[C_PITCH]) = make_uint2(lo, hi);
  }
}

// chroma sample of output pixel (x, y): jdsample.c fancy upsampling (dw x dh real samples in a plane of pitch `pitch`)
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int64_t pitch, int dw, int dh, int hs, int vs, int x, int y) {
  if (hs == 1) return (int)p[(int64_t)y * pitch + x];
  const int cx = x >> 1;
  if (vs == 1) {                                             // h2v1
    const uint8_t* row = p + (int64_t)y * pitch;
    if (dw <= 2) return (int)row[cx];
    const int a = (int)row[cx];
    if (x & 1) return cx == dw - 1 ? a : (a * 3 + (int)row[cx + 1] + 2) >> 2;
    return cx == 0 ? a : (a * 3 + (int)row[cx - 1] + 1) >> 2;
  }
  const int cy = y >> 1;                                     // h2v2
  if (dw <= 2) return (int)p[(int64_t)cy * pitch + cx];
  const int ny = (y & 1) ? (cy + 1 < dh ? cy + 1 : dh - 1) : (cy > 0 ? cy - 1 : 0);   // the context row: above for even, below for odd output rows
  const uint8_t* r0 = p + (int64_t)cy * pitch;
  const uint8_t* r1 = p + (int64_t)ny * pitch;
  const int cs = (int)r0[cx] * 3 + (int)r1[cx];
  if (x & 1) {
    if (cx == dw - 1) return (cs * 4 + 7) >> 4;
    return (cs * 3 + (int)r0[cx + 1] * 3 + (int)r1[cx + 1] + 7) >> 4;
  }
  if (cx == 0) return (cs * 4 + 8) >> 4;
  return (cs * 3 + (int)r0[cx - 1] * 3 + (int)r1[cx - 1] + 8) >> 4;
}
__device__ __forceinline__ unsigned clamp255(int v) { return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// grid.y = image, grid.x * 256 threads over its pixels
__global__ __launch_bounds__(256) void colour_kernel(const int64_t* __restrict__ imgs, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out) {
  const int64_t* im = imgs + (int64_t)blockIdx.y * IMG_WORDS;
  const int W = (int)im[I_W], H = (int)im[I_H];
  const int px = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (px >= W * H) return;
  const int y = px / W, x = px - y * W;
  const int64_t* c0 = im + I_COMP0;
  const int Y = (int)planes[c0[C_PLANE] + (int64_t)y * c0[C_PITCH] + x];
  uint8_t* o = out + im[I_OUT] + (int64_t)px * 3;
  if ((int)im[I_NCOMP] == 1) { o[0] = (uint8_t)Y; o[1] = (uint8_t)Y; o[2] = (uint8_t)Y; return; }
  const int hs = (int)im[I_HS], vs = (int)im[I_VS];
  const int dw = (W + hs - 1) / hs, dh = (H + vs - 1) / vs;
  const int64_t* c1 = c0 + 8;
  const int64_t* c2 = c0 + 16;
  const int cb = chroma_at(planes + c1[C_PLANE], c1[C_PITCH], dw, dh, hs, vs, x, y) - 128;
  const int cr = chroma_at(planes + c2[C_PLANE], c2[C_PITCH], dw, dh, hs, vs, x, y) - 128;
  // jdcolor.c build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768
  o[0] = (uint8_t)clamp255(Y + ((91881 * cr + 32768) >> 16));
  o[1] = (uint8_t)clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
  o[2] = (uint8_t)clamp255(Y + ((116130 * cb + 32768) >> 16));
}

}  // namespace jpeg
}  // namespace ssg

using namespace ssg;

// Decode a batch of baseline JPEG files whose marker segments the host has parsed (ssg_amd/jpeg.py builds every table below).
//   ecs          entropy-coded bytes of all files back to back (+ >= 32 zero bytes of padding at the end)
//   segs         int64 [nseg][5]: image, byte offset into ecs, byte length, first MCU, MCU count (one row per restart segment)
//   imgs         int64 [nimg][32]: W, H, components (1 | 3), luma sampling h, v, MCUs per row, MCU rows, byte offset into out, then per
//                component 8 words: first block in coef, blocks per row, block rows, byte offset into planes, plane pitch,
//                quantisation table index, DC table index, AC table index
//   look / maxcode / valoff / vals    derived Huffman tables (jdhuff.c jpeg_make_d_derived_tbl), qts uint16 [nqt][64] natural order
//   coef         workspace int16 [total blocks][64] (zeroed here), planes workspace uint8, out uint8 RGB pixels (H x W x 3 per image)
//   status       int32 [nimg] (zeroed here): non-zero = the file's entropy-coded data is damaged (bit 0: a zero run past coefficient 63,
//                bit 1: a segment ended before its MCUs were decoded); its pixels are then NOT what libjpeg produces -- the caller re-decodes
//                such files with the reference's decoder, which warns / raises as the reference does
extern "C" int ssg_jpeg_decode_batch(const uint8_t* ecs, const int64_t* segs, int nseg, const int64_t* imgs, int nimg, const uint16_t* look,
                                     const int32_t* maxcode, const int32_t* valoff, const uint8_t* vals, const uint16_t* qts, int16_t* coef,
                                     int64_t total_blocks, int max_blocks, uint8_t* planes, int max_pixels, uint8_t* out, int32_t* status, hipStream_t stream) {
  if (!status || nseg <= 0 || nimg <= 0 || total_blocks <= 0 || max_blocks <= 0 || max_pixels <= 0 || nimg > 21845) {
    ssg_set_error("ssg_jpeg_decode_batch: bad shape (nseg=%d nimg=%d blocks=%lld)", nseg, nimg, (long long)total_blocks);
    return SSG_ERR_INVALID;
  }
  SSG_HIP(hipMemsetAsync(coef, 0, (size_t)total_blocks * 64 * sizeof(int16_t), stream));
  SSG_HIP(hipMemsetAsync(status, 0, (size_t)nimg * sizeof(int32_t), stream));
  jpeg::Tables t{look, maxcode, valoff, vals};
  hipLaunchKernelGGL(jpeg::huffman_kernel, dim3((nseg + 63) / 64), dim3(64), 0, stream, ecs, segs, nseg, imgs, t, coef, status);
  hipLaunchKernelGGL(jpeg::idct_kernel, dim3((max_blocks + 63) / 64, nimg * 3), dim3(64), 0, stream, imgs, coef, qts, planes);
  hipLaunchKernelGGL(jpeg::colour_kernel, dim3((max_pixels + 255) / 256, nimg), dim3(256), 0, stream, imgs, planes, out);
  SSG_LAUNCH_CHECK("jpeg kernels");
  return SSG_OK;
}
