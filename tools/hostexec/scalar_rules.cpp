// The scalar numerics rules of the device code, executed on the host: csrc/ssg_common.h (half arithmetic as numpy does it, the one-rounding
// double -> half, final_dist from its compact form) and the scalar helpers of the kernels -- the integer half(sqrt(.)) rounding of
// gram_i8.hip, jaccard_scaled (jaccard.hip), numpy's pairwise summation (krecip.hip float32, cluster.hip float64 leaves), the rank key of
// the introsort replay (topk_intro.hip), the eps rule's surrogate bins (cluster.hip) -- are plain scalar C++ behind `__device__`.  The test
// cuts their text out of the .hip files into rules_cut.inc; with `__device__` mapped to host functions the SAME TEXT is compiled for x86 and
// compared with numpy on millions of inputs (tests/test_host_exec.py) -- a CPU-side pin of the rules DESIGN.md §4 lists, next to the
// `-m gpu` tests that run them on the GPU.  Test infrastructure: nothing here is linked into the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#undef __device__
#undef __forceinline__
#define __device__
#define __forceinline__ inline
static int g_sqrt_mode = 0;     // the device's v_sqrt_f32 is good to 1 ulp: the candidate may be off by one float either way
static inline float hx_sqrtf(float x) {
  const float r = sqrtf(x);
  return g_sqrt_mode > 0 ? nextafterf(r, INFINITY) : g_sqrt_mode < 0 ? nextafterf(r, 0.0f) : r;
}
#define __builtin_amdgcn_sqrtf hx_sqrtf
static inline long long hx_double_as_longlong(double x) { long long r; memcpy(&r, &x, 8); return r; }
#define __double_as_longlong hx_double_as_longlong
static inline double hx_longlong_as_double(long long x) { double r; memcpy(&r, &x, 8); return r; }
#define __longlong_as_double hx_longlong_as_double
static inline unsigned hx_float_as_uint(float x) { unsigned r; memcpy(&r, &x, 4); return r; }
static inline float hx_uint_as_float(unsigned x) { float r; memcpy(&r, &x, 4); return r; }
#define __float_as_uint hx_float_as_uint
#define __uint_as_float hx_uint_as_float
#define threadIdx hx_threadIdx
#define blockIdx hx_blockIdx
#define blockDim hx_blockDim
#define gridDim hx_gridDim
struct HxDim { unsigned x, y, z; };
static HxDim hx_threadIdx = {0, 0, 0}, hx_blockIdx = {0, 0, 0}, hx_blockDim = {1, 1, 1}, hx_gridDim = {1, 1, 1};   // ONE thread: grid-stride kernels cover everything
#undef __global__
#define __global__
#undef __launch_bounds__
#define __launch_bounds__(x)
#include "../../self-similarity-grouping_amd/csrc/ssg_common.h"
// single host thread: the atomics of the union-find kernels are plain operations
#define __hip_atomic_load(p, order, scope) (*(p))
static inline int hx_atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
static inline int hx_atomicCAS(int* p, int cmp, int v) { const int o = *p; if (o == cmp) *p = v; return o; }
#define atomicMin hx_atomicMin
#define atomicCAS hx_atomicCAS
// ... and a wave is that one thread (lane 0): a shuffle returns the lane's own value, so cc_hook_kernel's segmented min over the lanes of a
// run degenerates to m = k and every edge hooks by itself
#define __shfl_up(v, d, w) (v)
#define __shfl_down(v, d, w) (v)
#include <algorithm>
#include <vector>
using std::max;
using std::min;
namespace ssg {
#include "rules_cut.inc"
#include "preprocess_cut.inc"
#include "cc_cut.inc"
#include "split_cut.inc"
#include "pool_cut.inc"
}
using namespace ssg;

extern "C" {
void hx_d2h(const double* x, long n, uint16_t* out) { for (long i = 0; i < n; i++) out[i] = d2h(x[i]); }
void hx_f2h(const float* x, long n, uint16_t* out) { for (long i = 0; i < n; i++) out[i] = f2h(x[i]); }
void hx_h2f(const uint16_t* x, long n, float* out) { for (long i = 0; i < n; i++) out[i] = h2f(x[i]); }
void hx_binop(int op, const uint16_t* a, const uint16_t* b, long n, uint16_t* out) {
  for (long i = 0; i < n; i++) out[i] = op == 0 ? h_add(a[i], b[i]) : op == 1 ? h_sub(a[i], b[i]) : op == 2 ? h_mul(a[i], b[i]) : h_div(a[i], b[i]);
}
void hx_final_dist(const uint16_t* jp, const uint16_t* vi, const uint16_t* vk, double lam, long n, double* out) {
  for (long i = 0; i < n; i++) out[i] = final_dist_value(jp[i], vi[i], vk[i], lam);
}
void hx_sqrt48(const long long* u, long n, int mode, uint16_t* out) {
  g_sqrt_mode = mode;
  for (long i = 0; i < n; i++) out[i] = sqrt_units48_to_half(u[i]);
  g_sqrt_mode = 0;
}
void hx_jaccard_scaled(const uint16_t* t, uint16_t om, long n, uint16_t* out) { for (long i = 0; i < n; i++) out[i] = jaccard_scaled(t[i], om); }
float hx_pairwise_sum_f32(const float* a, int n) { return pairwise_sum_f32(a, n); }
double hx_pw_leaf_f64(const unsigned long long* keys, long long off, int n) { return pw_leaf<double>(keys, off, n); }
float hx_pw_leaf_f32(const unsigned long long* keys, long long off, int n) { return pw_leaf<float>(keys, off, n); }
void hx_norm_key(const uint32_t* raw, float fmx, long n, uint32_t* out) { for (long i = 0; i < n; i++) out[i] = norm_key(raw[i], fmx); }
void hx_sur_bin(const float* x, long n, int* out) { for (long i = 0; i < n; i++) out[i] = sur_bin(x[i]); }
float hx_sur_bin_upper(int b) { return sur_bin_upper(b); }
// bucket rule of the device-sized sample sort (cluster.hip ss_bucket): table = 1023 ascending splitters + ~0
void hx_ss_bucket(const unsigned long long* table, const unsigned long long* keys, long n, int* out) { for (long i = 0; i < n; i++) out[i] = ss_bucket<1024>(table, keys[i]); }
// the two resize kernels of preprocess.hip (grid-stride, one thread here): uint8 [B,h,w,3] -> float32 [B,3,H,W]
void hx_preprocess(const uint8_t* src, int B, int h, int w, int H, int W, const int32_t* xmin, const int32_t* xcnt, const int32_t* kkx, int ksx,
                   const int32_t* ymin, const int32_t* ycnt, const int32_t* kky, int ksy, const float* m, const float* sd, uint8_t* tmp, float* out) {
  resize_h_u8_kernel(src, tmp, B, h, w, W, xmin, xcnt, kkx, ksx);
  resize_v_normalize_kernel(tmp, out, B, h, H, W, ymin, ycnt, kky, ksy, m[0], m[1], m[2], sd[0], sd[1], sd[2]);
}
// ssg_dbscan_cc's launch sequence (cluster.hip dbscan_cc_impl) with the kernels' own text: per-index kernels run index by index, the
// grid-stride ones (union, border) as one thread over the edge list in the order given; the exclusive scan of the root flags (a wave-level
// kernel on the device) is restated here
void hx_dbscan_cc(const int32_t* cnt, const int32_t* edges, unsigned long long ne, int N, int min_samples, int64_t* labels) {
  std::vector<int> parent(N), lab(N);
  std::vector<int32_t> rootflag(N);
  std::vector<int64_t> rootid(N + 1);
  hx_blockDim.x = 1; hx_gridDim.x = 1; hx_threadIdx.x = 0;
  for (int i = 0; i < N; i++) { hx_blockIdx.x = (unsigned)i; cc_init_kernel(cnt, N, min_samples, parent.data(), lab.data()); }
  hx_blockIdx.x = 0;
  if (ne) { cc_hook_kernel(edges, ne, nullptr, cnt, min_samples, parent.data()); cc_union_kernel(edges, ne, nullptr, cnt, min_samples, parent.data()); }
  for (int i = 0; i < N; i++) { hx_blockIdx.x = (unsigned)i; cc_flatten_kernel(cnt, N, min_samples, parent.data(), rootflag.data()); }
  int64_t run = 0;
  for (int i = 0; i < N; i++) { rootid[i] = run; run += rootflag[i]; }
  hx_blockIdx.x = 0;
  if (ne) cc_border_kernel(edges, ne, nullptr, cnt, min_samples, parent.data(), rootid.data(), lab.data(), N);      // (labels the core points too)
  else for (int i = 0; i < N; i++) { hx_blockIdx.x = (unsigned)i; cc_label_core_kernel(cnt, N, min_samples, parent.data(), rootid.data(), lab.data()); }
  for (int i = 0; i < N; i++) { hx_blockIdx.x = (unsigned)i; cc_finalize_kernel(lab.data(), N, labels); }
  hx_blockIdx.x = 0;
}
// split-half format of the embedding (conv.hip): n4 groups of 4 floats -> hi / lo dword pairs, the non-finite test, and back
void hx_split(const float* in, long n4, uint32_t* hi, uint32_t* lo, float* dec, uint8_t* nonfinite) {
  for (long i = 0; i < n4; i++) {
    const float4 v = make_float4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
    uint2 h, l;
    split_encode4(v, h, l);
    hi[2 * i] = h.x; hi[2 * i + 1] = h.y; lo[2 * i] = l.x; lo[2 * i + 1] = l.y;
    nonfinite[i] = split_hi_nonfinite(h) ? 1 : 0;
    const float4 d = split_decode4(h, l);
    dec[4 * i] = d.x; dec[4 * i + 1] = d.y; dec[4 * i + 2] = d.z; dec[4 * i + 3] = d.w;
  }
}
// the non-GEMM layers of the embedding on split-half tensors (conv.hip; grid-stride, one thread here): n floats <-> h8l8, 3x3 / 2 max
// pooling, global + stripe average pooling
void hx_h8l8_encode(const float* in, float* out, long n) { h8l8_encode_kernel(in, out, n / 8, 1.f); }
void hx_h8l8_decode(const float* in, float* out, long n) { h8l8_decode_kernel(in, out, n / 8, 1.f); }
void hx_maxpool_h8l8(const float* in, float* out, int B, int H, int W, int C, int OH, int OW) { maxpool3x3s2_h8l8_kernel(in, out, B, H, W, C, OH, OW); }
void hx_gap_h8l8(const float* in, float* out, int B, int H, int W, int C, int S) { gap_stripes_h8l8_kernel(in, out, B, H, W, C, S); }
void hx_units24(const uint32_t* h, long n, long long* out) { for (long i = 0; i < n; i++) out[i] = half_units24(h[i]); }
}
