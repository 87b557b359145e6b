// Phase accounting of conv_dma_kernel (csrc/conv.hip built with SSG_DMA_PROF): s_memtime stamps per workgroup --
//   0 start | 1 first k-tile landed (prologue) | 2 main loop done | 3 tail DMAs drained | 4 epilogue done
// build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSSG_DMA_PROF -I self-similarity-grouping_amd/csrc tools/micro/conv_prof.hip -o /tmp/conv_prof && /tmp/conv_prof
// Shapes: layer3 conv3 + residual (128 x 256 tiles, K = 256), layer3 conv1 (256 x 256 tiles, K = 1024), layer3 3x3 (K = 2304).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ssg_api.hip"
#include "conv.hip"

static void fill_halves(std::vector<uint16_t>& v, unsigned seed, int emin, int espan) {
  unsigned s = seed;
  for (size_t i = 0; i < v.size(); i++) { s = s * 1664525u + 1013904223u; v[i] = (uint16_t)(((emin + (s >> 20) % espan) << 10) | ((s >> 8) & 0x3ff) | ((s & 1) << 15)); }
}

template <typename F>
static void run(const char* name, int B, int H, int W, int Cin, int Cout, int k, int pad, bool res, int BM, F launch) {
  using namespace ssg;
  const int M = B * H * W, Kpad = k * k * Cin;
  std::vector<uint16_t> hx((size_t)M * Cin * 2), hw((size_t)Cout * Kpad * 2), hr((size_t)M * Cout * 2);
  fill_halves(hx, 1, 8, 6); fill_halves(hw, 2, 6, 6); fill_halves(hr, 3, 8, 6);
  std::vector<float> ones(Cout, 1.f), zeros(Cout, 0.f);
  void *x, *w, *r, *out; float *cs, *bi; unsigned long long* prof; int* ovf;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&r, hr.size() * 2); hipMalloc(&out, hr.size() * 2);
  hipMalloc(&cs, Cout * 4); hipMalloc(&bi, Cout * 4); hipMalloc(&ovf, 4); hipMemset(ovf, 0, 4);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(r, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(cs, ones.data(), Cout * 4, hipMemcpyHostToDevice); hipMemcpy(bi, zeros.data(), Cout * 4, hipMemcpyHostToDevice);
  const int tiles = ((M + BM - 1) / BM) * (Cout / 256);
  hipMalloc(&prof, (size_t)tiles * 8 * 8);
  ConvParams p;
  p.in = (const float*)x; p.w = (const float*)w; p.bias = bi; p.res = res ? (const float*)r : nullptr; p.out = (float*)out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = H; p.OW = W; p.Cout = Cout; p.KH = k; p.KW = k; p.stride = 1; p.pad = pad; p.relu = 1;
  p.M = M; p.Kpad = Kpad; p.variant = 0; p.in_bytes = (unsigned)((size_t)M * Cin * 4);
  p.in2 = nullptr; p.H2 = p.W2 = p.Cin2 = 0; p.stride2 = 1; p.nk1 = Kpad / 16; p.in2_bytes = 0; p.rowterm = nullptr; p.epi = 0; p.tilemin = nullptr; p.tmin_ld = 0;
  p.out_split = 1; p.res_split = 1; p.acc_scale = 1.f; p.cscale = cs; p.overflow = ovf; p.products = 3; p.prof = prof;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0); launch(p, tiles); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  std::vector<unsigned long long> hp((size_t)tiles * 8);
  hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost);
  const char* names[4] = {"prologue (first tile lands)", "main loop", "tail DMA drain + barrier", "epilogue"};
  double tot = 0, ph[4] = {0, 0, 0, 0};
  for (int t = 0; t < tiles; t++) { for (int i = 0; i < 4; i++) ph[i] += (double)(hp[t * 8 + i + 1] - hp[t * 8 + i]); tot += (double)(hp[t * 8 + 4] - hp[t * 8]); }
  const double flop = 2.0 * M * Cout * (double)Kpad;
  printf("%s  B=%d M=%d K=%d N=%d tiles=%d (%d x 256): %.3f ms, %.1f TFLOP/s fp32-equivalent\n", name, B, M, Kpad, Cout, tiles, BM, best, flop / best / 1e9);
  for (int i = 0; i < 4; i++) printf("  %-30s %9.0f ticks/workgroup (%4.1f %%)\n", names[i], ph[i] / tiles, 100.0 * ph[i] / tot);
  printf("  %-30s %9.0f ticks/workgroup; MFMA floor of the main loop alone on its CU: %.0f cycles\n", "total", tot / tiles, (double)BM * 256 * Kpad * 2 * 3 / 4096.0);
  hipFree(x); hipFree(w); hipFree(r); hipFree(out); hipFree(cs); hipFree(bi); hipFree(prof); hipFree(ovf);
}

int main(int argc, char** argv) {
  using namespace ssg;
  const int B = argc > 1 ? atoi(argv[1]) : 1000;
  run("layer3 conv3 + res, 128x256", B, 16, 8, 256, 1024, 1, 0, true, 128, [](const ConvParams& p, int tiles) { hipLaunchKernelGGL(conv_dma_kernel<256>, dim3(tiles), dim3(512), 0, 0, p); });
  run("layer3 conv3 + res, 256x256", B, 16, 8, 256, 1024, 1, 0, true, 256, [](const ConvParams& p, int tiles) { hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4>), dim3(tiles), dim3(1024), 0, 0, p); });
  run("layer3 conv1, 256x256", B, 16, 8, 1024, 256, 1, 0, false, 256, [](const ConvParams& p, int tiles) { hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4>), dim3(tiles), dim3(1024), 0, 0, p); });
  run("layer3 3x3, 256x256", B, 16, 8, 256, 256, 3, 1, false, 256, [](const ConvParams& p, int tiles) { hipLaunchKernelGGL((conv_dma_kernel<256, false, 256, false, 4>), dim3(tiles), dim3(1024), 0, 0, p); });
  run("layer4 conv3 + res, 128x256", B, 8, 4, 512, 2048, 1, 0, true, 128, [](const ConvParams& p, int tiles) { hipLaunchKernelGGL(conv_dma_kernel<256>, dim3(tiles), dim3(512), 0, 0, p); });
  return 0;
}
