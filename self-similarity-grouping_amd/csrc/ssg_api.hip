// ssg_api.hip -- C-ABI plumbing shared by all stages: error reporting, version, host-side
// half conversion used for scalar parameters.
#include "ssg_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void ssg_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int ssg_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return SSG_OK;
  ssg_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return SSG_ERR_HIP;
}

extern "C" const char* ssg_last_error(void) { return g_err; }
extern "C" int ssg_version(void) { return 100; }   // round 1

// half(1 - lambda): the weak python scalar of `jaccard_dist*(1-lambda_value)` (rerank.py:122)
extern "C" uint16_t ssg_double_to_half_bits(double d) {
  uint64_t b; memcpy(&b, &d, 8);
  const uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
  const int e = (int)((b >> 52) & 0x7ff);
  uint64_t m = b & 0xfffffffffffffULL;
  if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
  if (e == 0) return sign;
  const int he = e - 1023 + 15;
  if (he >= 31) return (uint16_t)(sign | 0x7c00u);
  m |= 1ULL << 52;
  int shift = 42;
  if (he <= 0) { shift = 43 - he; if (shift > 63) return sign; }
  uint64_t q = m >> shift;
  const uint64_t rem = m & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
  if (rem > half || (rem == half && (q & 1))) q++;
  if (he <= 0) return (uint16_t)(sign | (uint16_t)q);
  uint32_t r = ((uint32_t)he << 10) + (uint32_t)(q - 0x400);
  if (r >= 0x7c00u) r = 0x7c00u;
  return (uint16_t)(sign | r);
}

// device self-test hooks used by the parity tests (exhaustive half-function checks)
namespace ssg {
__global__ void half_fn_table_kernel(int which, uint16_t* out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 65536u) return;
  const hbits x = (hbits)i;
  hbits r = 0;
  if (which == 0) r = d2h(exp((double)h2f(x)));                 // correctly rounded half exp(x)
  else if (which == 1) r = h_sub(H_ONE, h_exp_neg(x));          // 1 - exp(-x)            rerank.py:38
  else if (which == 2) r = h_mul(x, x);                         // np.power(half, 2)      rerank.py:62
  else if (which == 3) r = h_sub(H_ONE, h_div(x, h_sub(H_TWO, x)));   // 1 - t/(2-t)     rerank.py:115
  else if (which == 4) r = d2h(sqrt((double)h2f(x)));           // half(sqrt_f64)
  out[i] = r;
}
__global__ void half_binop_kernel(int which, const uint16_t* a, const uint16_t* b, int n, uint16_t* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  out[i] = which == 0 ? h_add(a[i], b[i]) : which == 1 ? h_div(a[i], b[i]) : which == 2 ? h_mul(a[i], b[i]) : h_sub(a[i], b[i]);
}
__global__ void d2h_kernel(const double* a, int n, uint16_t* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) out[i] = d2h(a[i]);
}
}  // namespace ssg

extern "C" int ssg_selftest_half_table(int which, uint16_t* out65536, hipStream_t stream) {
  hipLaunchKernelGGL(ssg::half_fn_table_kernel, dim3(256), dim3(256), 0, stream, which, out65536);
  SSG_LAUNCH_CHECK("half_fn_table_kernel");
  return SSG_OK;
}
extern "C" int ssg_selftest_half_binop(int which, const uint16_t* a, const uint16_t* b, int n, uint16_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(ssg::half_binop_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, which, a, b, n, out);
  SSG_LAUNCH_CHECK("half_binop_kernel");
  return SSG_OK;
}
extern "C" int ssg_selftest_d2h(const double* a, int n, uint16_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(ssg::d2h_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, n, out);
  SSG_LAUNCH_CHECK("d2h_kernel");
  return SSG_OK;
}
