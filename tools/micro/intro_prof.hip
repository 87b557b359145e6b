// Cycle accounting of the introsort ranking kernel (csrc/topk_intro.hip built with SSG_INTRO_PROF).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSG_INTRO_PROF -I self-similarity-grouping_amd/csrc tools/micro/intro_prof.hip -o /tmp/intro_prof && /tmp/intro_prof 16000
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ssg_api.hip"
#include "topk_intro.hip"

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16000;
  const int nrows = argc > 2 ? atoi(argv[2]) : N;
  const int K = 21;
  std::vector<uint16_t> h((size_t)nrows * N);
  unsigned s = 12345;
  for (size_t i = 0; i < h.size(); i++) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3400 + ((s >> 12) % 1500)); }   // ~1500 distinct keys: ties everywhere
  std::vector<uint32_t> rm(nrows, 0x3c00);
  uint16_t* D; uint32_t* rmax; int32_t* rank; void* ws = nullptr;
  hipMalloc(&D, h.size() * 2); hipMalloc(&rmax, nrows * 4); hipMalloc(&rank, (size_t)nrows * K * 4);
  hipMemcpy(D, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(rmax, rm.data(), nrows * 4, hipMemcpyHostToDevice);
  size_t wsb = ssg_topk_rank_introsort_ws_bytes(N, nrows);
  if (wsb) hipMalloc(&ws, wsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
#ifdef SSG_INTRO_PROF
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(ssg::intro::g_prof), zero, sizeof(zero));
#endif
    hipEventRecord(e0);
    int rc = ssg_topk_rank_introsort(D, rmax, N, nrows, K, rank, ws, wsb, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d N=%d rows=%d  %.3f ms\n", rc, N, nrows, ms);
#ifdef SSG_INTRO_PROF
    unsigned long long p[16];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(ssg::intro::g_prof), sizeof(p));
    const char* names_lds[16] = {"load", "A", "B", "C", "D", "E", "F", "-", "insertion", "sort_total(incl. above)", "-", "wA", "wB", "wC", "wD", "wE"};
    const char* names_str[16] = {"s:A+keyclass", "s:B1 masks", "s:B2 counts", "s:C/D", "s:E1 copy", "s:X gather", "s:Y scatter", "LDS stage (sort_prefix + hand-over)", "s:X rank search", "s:X loop", "s:Y rank search", "s:Y loop", "-", "-", "-", "-"};
    const char** names = (ssg_topk_rank_introsort_ws_bytes(N, nrows) > (size_t)nrows * 9000 ? names_str : names_lds);
    for (int i = 0; i < 16; i++) if (p[i]) printf("  %-24s %10.1f cycles/row\n", names[i], (double)p[i] / nrows);
#endif
  }
  return 0;
}
