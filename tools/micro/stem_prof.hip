// Phase accounting of the fused stem kernel (csrc/stem_pool.hip built with SSG_STEM_PROF).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_STEM_PROF -I self-similarity-grouping_amd/csrc tools/micro/stem_prof.hip -o /tmp/stem_prof && /tmp/stem_prof 512
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ssg_api.hip"
#include "stem_pool.hip"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 512, H = 256, W = 128;
  const size_t nimg = (size_t)B * 3 * H * W;
  std::vector<float> himg(nimg);
  unsigned s = 7;
  for (size_t i = 0; i < nimg; i++) { s = s * 1664525u + 1013904223u; himg[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  std::vector<uint16_t> hw((size_t)64 * 224 * 2);
  for (size_t i = 0; i < hw.size(); i++) { s = s * 1664525u + 1013904223u; hw[i] = (uint16_t)(((8 + (s >> 20) % 6) << 10) | ((s >> 8) & 0x3ff) | ((s & 1) << 15)); }
  std::vector<float> ones(64, 1.f), zeros(64, 0.f);
  float *img, *cs, *bi; void *w, *out; unsigned long long* prof;
  const int nwg = B * (H / 4);
  hipMalloc(&img, nimg * 4); hipMalloc(&w, hw.size() * 2); hipMalloc(&cs, 256); hipMalloc(&bi, 256); hipMalloc(&out, (size_t)B * 64 * 32 * 64 * 4);
  
  hipMemcpy(img, himg.data(), nimg * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(cs, ones.data(), 256, hipMemcpyHostToDevice); hipMemcpy(bi, zeros.data(), 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    int rc = ssg_stem_pool_nchw_x(img, 0, w, bi, cs, out, B, H, W, nullptr, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d B=%d workgroups=%d  %.3f ms\n", rc, B, nwg, ms);
  }
#ifdef SSG_STEM_PROF
  unsigned long long hp[8];
  hipMemcpyFromSymbol(hp, HIP_SYMBOL(ssg::stem::g_stem_prof), sizeof(hp));
  const char* names[4] = {"weights + first window", "implicit GEMM", "patch update + conv tile", "maxpool + store"};
  double tot = 0; for (int i = 0; i < 4; i++) tot += (double)hp[i];
  for (int i = 0; i < 4; i++) printf("  %-26s %6.1f %%\n", names[i], 100.0 * hp[i] / tot);
#endif
  return 0;
}
