// ssg_common.h -- shared device helpers for the SSG grouping kernels (gfx950 only).
//
// Half arithmetic follows numpy's half loops exactly (the reference computes the whole
// re-rank in np.float16, reid/rerank.py:33-122): every binary op is an IEEE float32 op
// followed by one round-to-nearest-even to half; double -> half is a single direct RNE
// (npy_double_to_half), never via float.
#pragma once
// numpy rounds a*b and +c separately (e.g. rerank.py:122 final_dist, the float64 distance epilogues): no FMA contraction
// anywhere in this library, whatever flags it is built with.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSG_OK 0
#define SSG_ERR_INVALID (-1)
#define SSG_ERR_HIP (-2)
#define SSG_ERR_OVERFLOW (-3)
#define SSG_ERR_NAN (-4)

namespace ssg {

typedef uint16_t hbits;  // IEEE binary16 bit pattern

__device__ __forceinline__ float h2f(hbits h) { return (float)__builtin_bit_cast(_Float16, h); }
// float -> half RNE (v_cvt_f16_f32; f16 denormals are kept in hipcc's default mode)
__device__ __forceinline__ hbits f2h(float f) { return __builtin_bit_cast(hbits, (_Float16)f); }

// double -> half, one rounding (bit-exact port of npy_double_to_half; gfx950 has no
// direct f64->f16 convert and f64->f32->f16 double-rounds).
__device__ __forceinline__ hbits d2h(double d) {
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  const hbits sign = (hbits)((b >> 48) & 0x8000u);
  const int e = (int)((b >> 52) & 0x7ff);
  uint64_t m = b & 0xfffffffffffffULL;
  if (e == 0x7ff) return (hbits)(sign | 0x7c00u | (m ? 0x200u : 0u));
  if (e == 0) return sign;
  const int he = e - 1023 + 15;
  if (he >= 31) return (hbits)(sign | 0x7c00u);
  m |= 1ULL << 52;
  int shift = 42;
  if (he <= 0) { shift = 43 - he; if (shift > 63) return sign; }
  uint64_t q = m >> shift;
  const uint64_t rem = m & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
  if (rem > half || (rem == half && (q & 1))) q++;
  if (he <= 0) return (hbits)(sign | (hbits)q);
  uint32_t r = ((uint32_t)he << 10) + (uint32_t)(q - 0x400);
  if (r >= 0x7c00u) r = 0x7c00u;
  return (hbits)(sign | r);
}

__device__ __forceinline__ hbits h_add(hbits a, hbits b) { return f2h(h2f(a) + h2f(b)); }
__device__ __forceinline__ hbits h_sub(hbits a, hbits b) { return f2h(h2f(a) - h2f(b)); }
__device__ __forceinline__ hbits h_mul(hbits a, hbits b) { return f2h(h2f(a) * h2f(b)); }
// IEEE-correct float division (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ hbits h_div(hbits a, hbits b) { return f2h(h2f(a) / h2f(b)); }
__device__ __forceinline__ bool h_isnan(hbits h) { return (h & 0x7fffu) > 0x7c00u; }
// correctly rounded half exp(-x): f64 exp (<=1 ulp) then one direct rounding
__device__ __forceinline__ hbits h_exp_neg(hbits x) { return d2h(exp(-(double)h2f(x))); }

constexpr hbits H_ONE = 0x3c00, H_TWO = 0x4000;

// final_dist[i,k] of reid/rerank.py:122 rebuilt from the compact representation:
//   f64(J'[i,k]) + f64(half(v_i + v_k)) * lambda      (J' = half(J * half(1-lambda)))
__device__ __forceinline__ double final_dist_value(hbits jp, hbits vi, hbits vk, double lambda_value) {
  return (double)h2f(jp) + (double)h2f(h_add(vk, vi)) * lambda_value;
}

// wave64 helpers
// value of lane (l ^ 1): a DPP quad permutation [1,0,3,2] on the way into the VALU -- __shfl_xor(v, 1) compiles to ds_bpermute_b32,
// an LDS-pipe round trip per dword (the split-half epilogues exchange four dwords per 16-byte store).  Every lane must be active.
__device__ __forceinline__ unsigned lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ULL << lane_id()) - 1ULL; }

}  // namespace ssg

// host-side error plumbing (ssg_api.cpp)
void ssg_set_error(const char* fmt, ...);
int ssg_check_hip(hipError_t e, const char* what);
#define SSG_HIP(call) do { int rc_ = ssg_check_hip((call), #call); if (rc_) return rc_; } while (0)
#define SSG_LAUNCH_CHECK(name) do { int rc_ = ssg_check_hip(hipGetLastError(), name); if (rc_) return rc_; } while (0)
