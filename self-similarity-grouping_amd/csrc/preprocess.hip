// preprocess.hip -- input side of the extraction loaders on the GPU.
//
// Replaces, for decoded RGB images, the per-image CPU transform of selftraining.py:43-47 applied by
// reid/utils/data/preprocessor.py:22-30:
//   Resize((H, W))  = PIL.Image.resize((W, H), BILINEAR): Pillow's separable resampling on 8-bit channels
//                     (libImaging/Resample.c: per output pixel a window of 22-bit fixed-point triangle-filter
//                     coefficients, accumulator started at 1 << 21, >> 22, clamped to 0..255; horizontal pass first,
//                     8-bit intermediate)
//   ToTensor()      = uint8 HWC -> float32 CHW / 255
//   Normalize(m, s) = (x - m[c]) / s[c]
// Bit-exact with Pillow: the coefficients are the same integers (computed in float64 on the host,
// ssg_amd/preprocessor.py), the arithmetic is the same int32 arithmetic, the float32 division is IEEE.
// Both passes are streaming kernels over a batch of equally sized images (HBM-bound, a few bytes per pixel).
#include "ssg_common.h"
#include <algorithm>

namespace ssg {

// tmp[b, y, X, c] = clip8((2^21 + sum_t src[b, y, xmin[X]+t, c] * kk[X, t]) >> 22)
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int B, int h, int w, int W,
                                                          const int32_t* __restrict__ xmin, const int32_t* __restrict__ xcnt,
                                                          const int32_t* __restrict__ kk, int ksize) {
  const int64_t total = (int64_t)B * h * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const int64_t by = i / W;                          // b * h + y
    const uint8_t* row = src + by * (int64_t)w * 3;
    const int lo = xmin[X], n = xcnt[X];
    const int32_t* k = kk + (int64_t)X * ksize;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int t = 0; t < n; t++) {
      const int c = k[t];
      const uint8_t* p = row + (int64_t)(lo + t) * 3;
      a0 += (int)p[0] * c; a1 += (int)p[1] * c; a2 += (int)p[2] * c;
    }
    uint8_t* o = tmp + i * 3;
    o[0] = (uint8_t)min(max(a0 >> 22, 0), 255); o[1] = (uint8_t)min(max(a1 >> 22, 0), 255); o[2] = (uint8_t)min(max(a2 >> 22, 0), 255);
  }
}

// out[b, c, Y, X] = (float(clip8(vertical pass of tmp)) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, int B, int h, int H, int W,
                                                                 const int32_t* __restrict__ ymin, const int32_t* __restrict__ ycnt,
                                                                 const int32_t* __restrict__ kk, int ksize, float m0, float m1, float m2, float s0,
                                                                 float s1, float s2) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const int Y = (int)((i / W) % H);
    const int64_t b = i / ((int64_t)W * H);
    const int lo = ymin[Y], n = ycnt[Y];
    const int32_t* k = kk + (int64_t)Y * ksize;
    const uint8_t* col = tmp + (b * h * (int64_t)W + X) * 3;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int t = 0; t < n; t++) {
      const int c = k[t];
      const uint8_t* p = col + (int64_t)(lo + t) * W * 3;
      a0 += (int)p[0] * c; a1 += (int)p[1] * c; a2 += (int)p[2] * c;
    }
    const float v0 = (float)min(max(a0 >> 22, 0), 255), v1 = (float)min(max(a1 >> 22, 0), 255), v2 = (float)min(max(a2 >> 22, 0), 255);
    const int64_t plane = (int64_t)H * W;
    float* o = out + b * 3 * plane + (int64_t)Y * W + X;
    o[0] = (v0 / 255.0f - m0) / s0;
    o[plane] = (v1 / 255.0f - m1) / s1;
    o[2 * plane] = (v2 / 255.0f - m2) / s2;
  }
}

__global__ __launch_bounds__(256) void clamp_sqrt_kernel(float* __restrict__ x, int64_t n, float lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    x[i] = sqrtf(v < lo ? lo : v);       // NaN stays NaN like torch.clamp
  }
}

}  // namespace ssg

using namespace ssg;

extern "C" int ssg_preprocess_u8(const uint8_t* src, int B, int h, int w, int H, int W, const int32_t* xmin, const int32_t* xcnt, const int32_t* xk,
                                 int xksize, const int32_t* ymin, const int32_t* ycnt, const int32_t* yk, int yksize, const float* mean3_host,
                                 const float* std3_host, uint8_t* tmp, float* out, hipStream_t stream) {
  if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || xksize <= 0 || yksize <= 0 || !mean3_host || !std3_host) {
    ssg_set_error("ssg_preprocess_u8: bad shape B=%d %dx%d -> %dx%d", B, h, w, H, W);
    return SSG_ERR_INVALID;
  }
  const int64_t n1 = (int64_t)B * h * W, n2 = (int64_t)B * H * W;
  const int g1 = (int)std::min<int64_t>((n1 + 255) / 256, 16384), g2 = (int)std::min<int64_t>((n2 + 255) / 256, 16384);
  hipLaunchKernelGGL(resize_h_u8_kernel, dim3(g1), dim3(256), 0, stream, src, tmp, B, h, w, W, xmin, xcnt, xk, xksize);
  hipLaunchKernelGGL(resize_v_normalize_kernel, dim3(g2), dim3(256), 0, stream, tmp, out, B, h, H, W, ymin, ycnt, yk, yksize, mean3_host[0],
                     mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
  SSG_LAUNCH_CHECK("preprocess kernels");
  return SSG_OK;
}

namespace ssg {
// backward of dist = sqrt(clamp(sq, min = lo)) w.r.t. sq's two arguments, folded into one symmetric weight matrix:
//   w[i,j] = sq[i,j] >= lo ? g[i,j] / dist[i,j] : 0      (clamp passes the gradient where its input is >= min; d sqrt = 1 / (2 sqrt))
//   S[i,j] = w[i,j] + w[j,i]   (x_i enters row i and column i of the distance matrix),  rowsum[i] = sum_j S[i,j]
// S is written with a row pitch of `ld` floats (zero padding up to the GEMM's K granule), one wave per row.
__global__ __launch_bounds__(256) void triplet_grad_weights_kernel(const float* __restrict__ g, const float* __restrict__ sq, const float* __restrict__ dist,
                                                                   int n, int ld, float lo, float* __restrict__ S, float* __restrict__ rowsum) {
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= n) return;
  const int lane = lane_id();
  float acc = 0.f;
  for (int j = lane; j < ld; j += 64) {
    float s = 0.f;
    if (j < n && j != i) {     // the diagonal multiplies x_i - x_i = 0: left out, so that a float32 residue in sq[i,i] (|x|^2 + |x|^2 - 2 x.x is not
                               // exactly 0) cannot put a huge weight 1 / dist[i,i] into rowsum and S x, where it would only cancel approximately
      const int64_t a = (int64_t)i * n + j, b = (int64_t)j * n + i;
      const float wij = sq[a] >= lo ? g[a] / dist[a] : 0.f;
      const float wji = sq[b] >= lo ? g[b] / dist[b] : 0.f;
      s = wij + wji;
    }
    S[(int64_t)i * ld + j] = s;
    acc += s;
  }
  for (int sh = 1; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh, 64);
  if (lane == 0) rowsum[i] = acc;
}
// grad_x[i,c] = rowsum[i] * x[i,c] - Sx[i,c]      (Sx with a row pitch of ldo floats)
__global__ __launch_bounds__(256) void triplet_grad_combine_kernel(const float* __restrict__ x, const float* __restrict__ rowsum, const float* __restrict__ Sx,
                                                                   int n, int d, int ldo, float* __restrict__ out) {
  const int64_t total = (int64_t)n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / d), c = (int)(t - (int64_t)i * d);
    out[t] = rowsum[i] * x[t] - Sx[(int64_t)i * ldo + c];
  }
}
}  // namespace ssg

// Backward of the TripletLoss pairwise block (reid/loss/triplet.py:28-31) -- the two elementwise halves around the fp32-MFMA GEMM S * x
// (ssg_conv2d_nhwc_f32 as a 1x1 convolution): grad_x = diag(rowsum(S)) x - S x with S = W + W^T, W = grad_dist / dist where the clamp
// passes the gradient.  ssg_triplet_grad_weights: g / sq / dist [n,n] -> S [n, ld] (ld >= n, zero padded), rowsum [n];
// ssg_triplet_grad_combine: x [n,d], Sx [n, ldo] -> grad_x [n,d].
extern "C" int ssg_triplet_grad_weights(const float* grad_dist, const float* sq, const float* dist, int n, int ld, float lo, float* S, float* rowsum,
                                        hipStream_t stream) {
  if (!grad_dist || !sq || !dist || !S || !rowsum || n <= 0 || ld < n) { ssg_set_error("ssg_triplet_grad_weights: bad arguments (n=%d ld=%d)", n, ld); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(ssg::triplet_grad_weights_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, grad_dist, sq, dist, n, ld, lo, S, rowsum);
  SSG_LAUNCH_CHECK("triplet_grad_weights_kernel");
  return SSG_OK;
}
extern "C" int ssg_triplet_grad_combine(const float* x, const float* rowsum, const float* Sx, int n, int d, int ldo, float* grad_x, hipStream_t stream) {
  if (!x || !rowsum || !Sx || !grad_x || n <= 0 || d <= 0 || ldo < d) { ssg_set_error("ssg_triplet_grad_combine: bad arguments (n=%d d=%d ldo=%d)", n, d, ldo); return SSG_ERR_INVALID; }
  const int64_t total = (int64_t)n * d;
  hipLaunchKernelGGL(ssg::triplet_grad_combine_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, stream, x, rowsum, Sx, n, d, ldo, grad_x);
  SSG_LAUNCH_CHECK("triplet_grad_combine_kernel");
  return SSG_OK;
}

extern "C" int ssg_clamp_sqrt_f32(float* x, int64_t n, float lo, hipStream_t stream) {
  if (n <= 0) return SSG_OK;
  hipLaunchKernelGGL(clamp_sqrt_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, x, n, lo);
  SSG_LAUNCH_CHECK("clamp_sqrt_kernel");
  return SSG_OK;
}
