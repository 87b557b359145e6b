// Micro-benchmark: HBM read bandwidth of the access patterns used by the N x N half-matrix passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void wave_per_row(const uint4* __restrict__ M, int N8, int nrows, unsigned* out, int pre) {
  for (int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); row < nrows; row += (int)((gridDim.x * blockDim.x) >> 6)) {
    const int lane = threadIdx.x & 63;
    const uint4* p = M + (int64_t)row * N8;
    unsigned acc = 0;
    if (pre == 1) { for (int c = lane; c < N8; c += 64) { uint4 x = p[c]; acc += x.x ^ x.y ^ x.z ^ x.w; } }
    else { for (int c = lane; c < N8; c += 256) { uint4 a = p[c], b = c + 64 < N8 ? p[c + 64] : a, d = c + 128 < N8 ? p[c + 128] : a, e = c + 192 < N8 ? p[c + 192] : a;
             acc += a.x ^ b.y ^ d.z ^ e.w ^ a.y ^ b.x ^ d.w ^ e.z; } }
    if (acc == 0x12345678u) out[row] = acc;
  }
}
__global__ void flat(const uint4* __restrict__ M, int64_t n, unsigned* out) {
  unsigned acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { uint4 x = M[i]; acc += x.x ^ x.y ^ x.z ^ x.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void fill_rows(uint4* __restrict__ M, int N8, int nrows) {
  const uint4 c = make_uint4(0x39993999u, 0x39993999u, 0x39993999u, 0x39993999u);
  for (int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); row < nrows; row += (int)((gridDim.x * blockDim.x) >> 6)) {
    const int lane = threadIdx.x & 63;
    uint4* p = M + (int64_t)row * N8;
    for (int c0 = lane; c0 < N8; c0 += 64) p[c0] = c;
  }
}
__global__ void fill_flat(uint4* __restrict__ M, int64_t n) {
  const uint4 c = make_uint4(0x39993999u, 0x39993999u, 0x39993999u, 0x39993999u);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) M[i] = c;
}
int main() {
  const int N = 16000, nrows = 16000; const int N8 = N / 8;
  uint4* M; unsigned* out; hipMalloc(&M, (size_t)nrows * N8 * 16); hipMalloc(&out, nrows * 4); hipMemset(M, 1, (size_t)nrows * N8 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)nrows * N8 * 16 / 1e9;
  for (int variant = 0; variant < 6; variant++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(wave_per_row, dim3(4000), dim3(256), 0, 0, M, N8, nrows, out, 1);
      if (variant == 1) hipLaunchKernelGGL(wave_per_row, dim3(4000), dim3(256), 0, 0, M, N8, nrows, out, 4);
      if (variant == 2) hipLaunchKernelGGL(wave_per_row, dim3(1280), dim3(256), 0, 0, M, N8, nrows, out, 1);
      if (variant == 3) hipLaunchKernelGGL(wave_per_row, dim3(2048), dim3(256), 0, 0, M, N8, nrows, out, 4);
      if (variant == 4) hipLaunchKernelGGL(flat, dim3(2048), dim3(256), 0, 0, M, (int64_t)nrows * N8, out);
      if (variant == 5) hipLaunchKernelGGL(flat, dim3(8192), dim3(256), 0, 0, M, (int64_t)nrows * N8, out);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const char* names[] = {"wave/row grid4000 1-load", "wave/row grid4000 4-loads", "wave/row grid1280 1-load", "wave/row grid2048 4-loads", "flat grid2048", "flat grid8192"};
    printf("%-28s %.3f ms  %.0f GB/s\n", names[variant], best, gb / best * 1e3);
  }
  for (int variant = 0; variant < 4; variant++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(fill_rows, dim3(4000), dim3(256), 0, 0, M, N8, nrows);
      if (variant == 1) hipLaunchKernelGGL(fill_rows, dim3(1280), dim3(256), 0, 0, M, N8, nrows);
      if (variant == 2) hipLaunchKernelGGL(fill_flat, dim3(2048), dim3(256), 0, 0, M, (int64_t)nrows * N8);
      if (variant == 3) hipMemsetAsync(M, 1, (size_t)nrows * N8 * 16, 0);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const char* names[] = {"fill wave/row grid4000", "fill wave/row grid1280", "fill flat grid2048", "hipMemsetAsync"};
    printf("%-28s %.3f ms  %.0f GB/s\n", names[variant], best, gb / best * 1e3);
  }
  return 0;
}
