// Development check (round 5): is the LDS packed-half atomic add (ds_pk_add_rtn_f16) the same function as the vector unit's half add
// (what ssg_common.h h_add compiles to = numpy's half + half) -- rounding, denormal inputs and results, overflow to inf -- and does adding
// +0.0 leave the other half of the dword alone?  All 65536 x 65536 pairs of (accumulator, addend) on the low half; the high half holds a
// finite pattern.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pkadd_check.hip -o tools/micro/pkadd_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t hadd(uint16_t a, uint16_t b) {
  return __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, a) + (float)__builtin_bit_cast(_Float16, b)));
}
__device__ __forceinline__ bool is_nan(uint16_t h) { return (h & 0x7c00u) == 0x7c00u && (h & 0x3ffu); }
__global__ __launch_bounds__(256) void check(unsigned long long* bad /* [8] */, unsigned* first_bad /* [4] */) {
  __shared__ unsigned s[256];
  const int t = threadIdx.x;
  unsigned long long nb_pos = 0, nb_any = 0, nb_hi = 0, nb_old = 0;
  for (unsigned a = blockIdx.x; a < 65536u; a += gridDim.x) {
    for (unsigned b0 = 0; b0 < 65536u; b0 += 256) {
      const uint16_t b = (uint16_t)(b0 + t), acc = (uint16_t)a;
      uint16_t hi = (uint16_t)((a * 40503u + b * 2654435761u) >> 7);
      if ((hi & 0x7c00u) == 0x7c00u) hi &= 0x3fffu;                       // finite pattern in the other half
      for (int side = 0; side < 2; side++) {
        s[t] = side == 0 ? ((unsigned)acc | ((unsigned)hi << 16)) : ((unsigned)hi | ((unsigned)acc << 16));
        const unsigned vv = side == 0 ? (unsigned)b : ((unsigned)b << 16);
        const half2_t o = __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(&s[t]), __builtin_bit_cast(half2_t, vv));
        const unsigned old = __builtin_bit_cast(unsigned, o), now = s[t];
        const uint16_t got = side == 0 ? (uint16_t)now : (uint16_t)(now >> 16), other = side == 0 ? (uint16_t)(now >> 16) : (uint16_t)now;
        const uint16_t want = hadd(acc, b);
        const bool same = got == want || (is_nan(got) && is_nan(want));
        const bool nonneg = !(acc & 0x8000u) && !(b & 0x8000u) && !is_nan(acc) && !is_nan(b);
        if (!same) { nb_any++; if (nonneg) { nb_pos++; if (atomicAdd(&first_bad[3], 1u) == 0) { first_bad[0] = acc; first_bad[1] = b; first_bad[2] = got; } } }
        if (other != hi) nb_hi++;
        if (old != (side == 0 ? ((unsigned)acc | ((unsigned)hi << 16)) : ((unsigned)hi | ((unsigned)acc << 16)))) nb_old++;
      }
    }
  }
  atomicAdd(&bad[0], nb_any); atomicAdd(&bad[1], nb_pos); atomicAdd(&bad[2], nb_hi); atomicAdd(&bad[3], nb_old);
}
int main() {
  unsigned long long* bad; unsigned* fb;
  hipMalloc(&bad, 64); hipMalloc(&fb, 16); hipMemset(bad, 0, 64); hipMemset(fb, 0, 16);
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, fb);
  unsigned long long h[8]; unsigned f[4];
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 16, hipMemcpyDeviceToHost);
  printf("ds_pk_add_rtn_f16 vs v_add_f16 over 2 x 2^32 (accumulator, addend) pairs: %llu differ (any sign / NaN), %llu differ among non-negative non-NaN operands, "
         "%llu times the other half changed, %llu wrong returned old values\n", h[0], h[1], h[2], h[3]);
  if (h[1]) printf("first non-negative mismatch: acc %04x + b %04x -> atomic %04x\n", f[0], f[1], f[2]);
  return 0;
}
