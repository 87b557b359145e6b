// Phase accounting of the fused bottleneck kernel (csrc/bottleneck.hip built with SSG_BN_PROF).
// build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_BN_PROF -I self-similarity-grouping_amd/csrc tools/micro/bneck_prof.hip -o /tmp/bneck_prof && /tmp/bneck_prof 512
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ssg_api.hip"
#include "bottleneck.hip"

static void fill_halves(std::vector<uint16_t>& v, unsigned seed, int emin, int espan) {
  unsigned s = seed;
  for (size_t i = 0; i < v.size(); i++) { s = s * 1664525u + 1013904223u; v[i] = (uint16_t)(((emin + (s >> 20) % espan) << 10) | ((s >> 8) & 0x3ff) | ((s & 1) << 15)); }
}

int main(int argc, char** argv) {
  using namespace ssg::bneck;
#ifdef BN_LAYER2
  constexpr int KC = 512, KMID = 128, KIW = 16, KTH = 8, KNW = 8;
  const int B = argc > 1 ? atoi(argv[1]) : 512, H = 32, W = 16, C = 512, MID = 128;
#else
  constexpr int KC = 256, KMID = 64, KIW = 32, KTH = 4, KNW = 4;
  const int B = argc > 1 ? atoi(argv[1]) : 512, H = 64, W = 32, C = 256, MID = 64;
#endif
  const size_t nx = (size_t)B * H * W * C;
  std::vector<uint16_t> hx(nx * 2), hw1((size_t)MID * C * 2), hw2((size_t)MID * 9 * MID * 2), hw3((size_t)C * MID * 2);
  fill_halves(hx, 1, 8, 6); fill_halves(hw1, 2, 6, 6); fill_halves(hw2, 3, 6, 6); fill_halves(hw3, 4, 6, 6);
  std::vector<float> ones(C, 1.f), zeros(C, 0.f);
  void *x, *out, *w1, *w2, *w3; float *cs, *bi; unsigned long long* prof;
  hipMalloc(&x, nx * 4); hipMalloc(&out, nx * 4); hipMalloc(&w1, hw1.size() * 2); hipMalloc(&w2, hw2.size() * 2); hipMalloc(&w3, hw3.size() * 2);
  hipMalloc(&cs, C * 4); hipMalloc(&bi, C * 4);
  const int ntiles = B * (H / KTH);
  hipMalloc(&prof, (size_t)ntiles * 8 * 8);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(cs, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(bi, zeros.data(), C * 4, hipMemcpyHostToDevice);
  Params p;
  p.x = (const float*)x; p.out = (float*)out; p.w1 = (const float*)w1; p.b1 = bi; p.cs1 = cs; p.w2 = (const float*)w2; p.b2 = bi; p.cs2 = cs;
  p.w3 = (const float*)w3; p.b3 = bi; p.cs3 = cs; p.B = B; p.H = H; p.overflow = nullptr; p.prof = prof;
  using K = Cfg<KC, KMID, KIW, KTH, KC, KNW>;
  hipFuncSetAttribute((const void*)bottleneck_kernel<KC, KMID, KIW, KTH, KC, KNW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((bottleneck_kernel<KC, KMID, KIW, KTH, KC, KNW>), dim3(ntiles), dim3(KNW * 64), K::LDS, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d tiles=%d LDS=%d  %.3f ms  (%.2f TB/s x+out)\n", B, ntiles, K::LDS, ms, 2.0 * nx * 4 / ms / 1e9);
  }
#ifdef SSG_BN_PROF
  std::vector<unsigned long long> hp((size_t)ntiles * 8);
  hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost);
  const char* names[5] = {"phase 1 (conv1 loop)", "y1 write + W2 prologue", "phase 2 (conv2 loop)", "y2 write + phase 3 MMA", "epilogue (res + store)"};
  double tot = 0, ph[5] = {0, 0, 0, 0, 0};
  for (int t = 0; t < ntiles; t++) { for (int i = 0; i < 5; i++) ph[i] += (double)(hp[t * 8 + i + 1] - hp[t * 8 + i]); tot += (double)(hp[t * 8 + 5] - hp[t * 8]); }
  for (int i = 0; i < 5; i++) printf("  %-26s %9.0f ticks/workgroup (%4.1f %%)\n", names[i], ph[i] / ntiles, 100.0 * ph[i] / tot);
  printf("  %-26s %9.0f ticks/workgroup\n", "total", tot / ntiles);
#endif
  return 0;
}
