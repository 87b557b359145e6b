// What the matrix cores sustain on this part for the split-half multiply, with NOTHING else going on: the inner loop of the convolution
// kernels (three v_mfma_f32_32x32x16_f16 products per k-step into four 32 x 32 accumulator tiles) on register-resident operands --
// no LDS, no global memory, no barriers -- on every SIMD of the chip, for random operands and for zeros.  The chip clocks to its power
// budget (MI355X_MICROARCH.md, "DVFS give-back"): this is the ceiling a power-limited kernel can approach, next to the nominal
// 2516.6 TFLOP/s (2.4 GHz) bench.py prices against.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ seed, int iters, float* __restrict__ sink) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  v8h ah[2], al[2], bh[2], bl[2];
  {
    const uint4 s0 = seed[(t * 8 + 0) & 65535], s1 = seed[(t * 8 + 1) & 65535], s2 = seed[(t * 8 + 2) & 65535], s3 = seed[(t * 8 + 3) & 65535];
    const uint4 s4 = seed[(t * 8 + 4) & 65535], s5 = seed[(t * 8 + 5) & 65535], s6 = seed[(t * 8 + 6) & 65535], s7 = seed[(t * 8 + 7) & 65535];
    ah[0] = __builtin_bit_cast(v8h, s0); ah[1] = __builtin_bit_cast(v8h, s1); al[0] = __builtin_bit_cast(v8h, s2); al[1] = __builtin_bit_cast(v8h, s3);
    bh[0] = __builtin_bit_cast(v8h, s4); bh[1] = __builtin_bit_cast(v8h, s5); bl[0] = __builtin_bit_cast(v8h, s6); bl[1] = __builtin_bit_cast(v8h, s7);
  }
  v16f acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  std::vector<uint16_t> h(65536 * 8);
  uint4* seed; float* sink; hipMalloc(&seed, 65536 * 16); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 8; mode++) {
    unsigned s = 7u;
    // per thread 8 vectors of 8 halves: [ah0 ah1 al0 al1 bh0 bh1 bl0 bl1] (a = pixels, b = weights; h / l = hi / lo halves)
    // mode 0: random halves in [2^-6, 2) with random signs; 1: half of ALL elements zero; 2: zeros; 3: half of the PIXEL elements zero (post-ReLU,
    // hi and lo together), weights dense -- the convolutions' operand statistics; 4-6: as 3 with the LO halves cut to 8 / 5 / 0 significant bits
    // (what a coarser low part would buy in clock); 7: as 3 with a quarter of the pixel elements non-zero
    for (size_t e = 0; e < h.size(); e++) {
      s = s * 1664525u + 1013904223u;
      uint16_t r = (uint16_t)((((s >> 20) % 7 + 9) << 10) | ((s >> 8) & 0x3ff) | ((s & 1) << 15));
      const int vec = (int)((e / 8) % 8), elem = (int)(e % 8);
      const bool pixel = vec < 4, lo = (vec & 2) != 0;
      // the zero pattern of a pixel element must be the same in its hi and lo vector: derive it from (thread, vec & 1, elem)
      const unsigned zkey = (unsigned)((e / 64) * 16 + (vec & 1) * 8 + elem) * 2654435761u;
      if (mode == 1) r = ((s >> 3) & 1) ? 0 : r;
      else if (mode == 2) r = 0;
      else if (mode >= 3) {
        if (pixel && ((zkey >> 13) & (mode == 7 ? 3 : 1))) r = 0;
        if (lo && mode == 4) r &= 0xfff8;
        if (lo && mode == 5) r &= 0xffc0;
        if (lo && mode == 6) r = 0;
      }
      h[e] = r;
    }
    hipMemcpy(seed, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wps = 4; wps >= 4; wps /= 2) {
      const int wgs = 256 * wps, iters = 40000 / wps;       // 256 CUs x wps workgroups of 4 waves: wps waves per SIMD
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(wgs), dim3(256), 0, 0, seed, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = std::min(best, ms);
      }
      const double flop = (double)wgs * 4 * iters * 12 * 32768.0;
      const char* names[8] = {"random halves", "half of ALL elements zero", "all zeros", "pixels half zero, weights dense", "  + lo halves 8 significant bits", "  + lo halves 5 significant bits", "  + lo halves zero", "pixels 3/4 zero, weights dense"};
      printf("%-34s %d wave(s) per SIMD: %.2f ms -> %.0f TFLOP/s fp16 MFMA = %.0f fp32-equivalent (3 products); %.3f of 2516.6; implied clock %.2f GHz\n", names[mode], wps, best,
             flop / best / 1e9, flop / best / 1e9 / 3, flop / best / 1e9 / 2516.6, flop / best / 1e9 / 2516.6 * 2.4);
    }
  }
  return 0;
}
