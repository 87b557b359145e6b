// Ablation timing of the exact int8 Gram kernel (csrc/gram_i8.hip): -DSSG_GI_NO_EPI / -DSSG_GI_NO_LOADS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DSSG_GI_NO_EPI] -I self-similarity-grouping_amd/csrc tools/micro/gram_prof.hip -o /tmp/gram_prof && /tmp/gram_prof 16000 2048
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "ssg_api.hip"
#include "pairwise.hip"
#include "gram_i8.hip"

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16000, d = argc > 2 ? atoi(argv[2]) : 2048;
  std::vector<float> hx((size_t)N * d);
  unsigned s = 3;
  for (int i = 0; i < N; i++) {
    double nrm = 0;
    for (int k = 0; k < d; k++) { s = s * 1664525u + 1013904223u; const float v = ((s >> 8) & 0xffff) / 32768.f - 1.f; hx[(size_t)i * d + k] = v; nrm += (double)v * v; }
    const float inv = (float)(1.0 / std::sqrt(nrm));
    for (int k = 0; k < d; k++) hx[(size_t)i * d + k] *= inv;
  }
  float* x; void* E; int64_t* norms; int32_t* flag; uint16_t* D; uint32_t* rowmax;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&E, ssg_gram_i8_encoded_bytes(N, d, 3)); hipMalloc(&norms, (size_t)N * 8); hipMalloc(&flag, 4);
  hipMalloc(&D, (size_t)N * N * 2); hipMalloc(&rowmax, (size_t)N * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemset(flag, 0, 4);
  int rc = ssg_gram_i8_encode(x, N, d, 3, E, norms, flag, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    rc |= ssg_sqdist_self_i8(E, norms, N, d, 3, 0, N, 0, D, rowmax, flag, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d N=%d d=%d  %.3f ms  (%.0f TOP/s of 5033)\n", rc, N, d, ms, 9.0 * N * (double)N * d / ms / 1e9);
  }
  return 0;
}
