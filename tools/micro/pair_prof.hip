// Phase accounting of conv_pair_kernel (csrc/conv_pair.hip built with SSG_PAIR_PROF): cycle-counter stamps of thread 0 per workgroup, summed over
// the eight 128-channel chunks -- GEMM1 (conv3, K = 256), the chunk epilogues (scale, bias, residual, ReLU, re-encode, store + stash),
// GEMM2 (the next conv1's share of the chunk, K = 128), the last epilogue (y1n).
// build + run on the GPU box (the ablation switches give wrong results on purpose):
//   for v in "" -DSSG_PAIR_ABL_NOEPI -DSSG_PAIR_ABL_NOMMA "-DSSG_PAIR_ABL_NOEPI -DSSG_PAIR_ABL_NOMMA"; do
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_PAIR_PROF $v -I self-similarity-grouping_amd/csrc -I include tools/micro/pair_prof.hip -o /tmp/pair_prof && /tmp/pair_prof; done
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ssg_api.hip"
#include "conv.hip"
#include "conv_pair.hip"

static void fill_halves(std::vector<uint16_t>& v, unsigned seed, int emin, int espan, bool half_zero) {
  unsigned s = seed;
  for (size_t i = 0; i < v.size(); i++) {
    s = s * 1664525u + 1013904223u;
    v[i] = (uint16_t)(((emin + (s >> 20) % espan) << 10) | ((s >> 8) & 0x3ff));
    if (half_zero && ((s >> 5) & 1)) v[i] = 0;         // pixels after a ReLU
  }
}

int main(int argc, char** argv) {
  using namespace ssg;
  using namespace ssg::pairk;
  const int B = argc > 1 ? atoi(argv[1]) : 1000;
  const int M = B * 128, K1 = 256, C = 1024, N2 = 256;
  std::vector<uint16_t> hy((size_t)M * K1 * 2), hr((size_t)M * C * 2), hw3((size_t)C * K1 * 2), hw1((size_t)N2 * C * 2);
  fill_halves(hy, 1, 8, 6, true); fill_halves(hr, 3, 8, 6, true); fill_halves(hw3, 2, 4, 5, false); fill_halves(hw1, 4, 3, 5, false);
  std::vector<float> ones(C, 1.f), zeros(C, 0.f);
  void *y2, *res, *w3, *w1, *out, *y1n; float *cs, *bi; unsigned long long* prof; int* ovf;
  hipMalloc(&y2, hy.size() * 2); hipMalloc(&res, hr.size() * 2); hipMalloc(&out, hr.size() * 2); hipMalloc(&y1n, (size_t)M * N2 * 4);
  hipMalloc(&w3, hw3.size() * 2); hipMalloc(&w1, hw1.size() * 2); hipMalloc(&cs, C * 4); hipMalloc(&bi, C * 4); hipMalloc(&ovf, 4); hipMemset(ovf, 0, 4);
  hipMemcpy(y2, hy.data(), hy.size() * 2, hipMemcpyHostToDevice); hipMemcpy(res, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(cs, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(bi, zeros.data(), C * 4, hipMemcpyHostToDevice);
  const int tiles = (M + 127) / 128;
  hipMalloc(&prof, (size_t)tiles * 8 * 8); hipMemset(prof, 0, (size_t)tiles * 8 * 8);
  PairParams p;
  p.y2 = (const float*)y2; p.w3 = (const float*)w3; p.b3 = bi; p.cs3 = cs; p.res = (const float*)res; p.out = (float*)out;
  p.w1n = (const float*)w1; p.b1n = bi; p.cs1n = cs; p.y1n = (float*)y1n; p.M = M; p.overflow = ovf; p.prof = prof;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL((conv_pair_kernel<256, 1024, 256>), dim3(tiles), dim3(512), 0, 0, p); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  std::vector<unsigned long long> hp((size_t)tiles * 8);
  hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost);
  const char* names[5] = {"total", "GEMM1 (conv3, 16 k-tiles x 8 chunks)", "chunk epilogues (x 8)", "GEMM2 (next conv1, 8 k-tiles x 8 chunks)", "last epilogue (y1n)"};
  double ph[5] = {0, 0, 0, 0, 0};
  for (int t = 0; t < tiles; t++) for (int i = 0; i < 5; i++) ph[i] += (double)hp[(size_t)t * 8 + i];
  const char* var =
#if defined(SSG_PAIR_ABL_NOEPI) && defined(SSG_PAIR_ABL_NOMMA)
      "ABLATION: no chunk epilogue, no multiply (the k-tile stream + barriers alone)";
#elif defined(SSG_PAIR_ABL_NOEPI)
      "ABLATION: no chunk epilogue (no residual read, no store of out, stale stash)";
#elif defined(SSG_PAIR_ABL_NOMMA)
      "ABLATION: no multiply (k-tile stream, barriers, epilogues)";
#elif defined(SSG_PAIR_ABL_NORES)
      "ABLATION: no residual read (stores of out kept)";
#elif defined(SSG_PAIR_ABL_NOSTORE)
      "ABLATION: no store of out (residual read kept)";
#elif SSG_PAIR_AUX != 0
      "residual loads + stores of out with the nt cache policy";
#else
      "the kernel as shipped";
#endif
  const double flop = 2.0 * M * (double)C * (K1 + N2);
  printf("conv_pair_kernel [%s]  B=%d M=%d tiles=%d: %.3f ms, %.1f TFLOP/s fp32-equivalent\n", var, B, M, tiles, best, flop / best / 1e9);
  for (int i = 0; i < 5; i++) printf("  %-44s %9.0f ticks/workgroup (%4.1f %%)\n", names[i], ph[i] / tiles, 100.0 * ph[i] / ph[0]);
  printf("  MFMA floor per workgroup alone on its CU: GEMM1 %.0f + GEMM2 %.0f cycles\n", 128.0 * C * K1 * 2 * 3 / 4096.0, 128.0 * N2 * C * 2 * 3 / 4096.0);
  return 0;
}
