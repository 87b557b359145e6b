// Micro-benchmark: what the LAYOUT of an activation tensor costs the 1x1 convolutions of layer3 / layer4 in HBM efficiency.
// The LDS-DMA GEMM (csrc/conv.hip conv_dma_kernel) streams its pixel operand as k-tiles of 16 channels: one DMA instruction of a wave
// = 16 pixel rows x 64 bytes, the rows one whole NHWC pixel apart (K = 1024 channels: 4 KB).  Every access of the tile walk therefore
// opens another DRAM page; the epilogue writes 8 rows x 128 bytes per store instruction at the same 4 KB pitch.  This file runs the bare
// access patterns (no multiply) over 0.52 GB in three layouts:
//   0  row-major NHWC              [m][K * 4 B]                                 (what the tensors are today)
//   1  k-tile-major per 256 rows   [m / 256][k / 16][256 rows][64 B]            (a GEMM tile's k-tile is ONE contiguous 16 KB piece)
//   2  32 x 32 blocks              [m / 32][k / 32][32 rows][128 B]             (a wave's epilogue patch is one contiguous 4 KB piece)
// build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/tile_stream.hip -o /tmp/tile_stream && /tmp/tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>

#define LDSP(ptr_) ((__attribute__((address_space(3))) void*)(ptr_))

// (a function of its own: with the builtin called directly from the kernel template the HOST pass of hipcc drops the kernel's launch stub)
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, unsigned lds_addr, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, voff, 0, 0, 0);
}

// reader: one workgroup of NW waves per BM = 16 * NW row tile, NS stages in flight, k-tiles of 64 bytes per row
template <int MODE, int NW, int NS>
__global__ __launch_bounds__(NW * 64) void read_tiles(const unsigned char* A, int M, int nk, unsigned bytes, unsigned* sink, int shared) {
  constexpr int BM = NW * 16;
  __shared__ __attribute__((aligned(1024))) unsigned char st[NS][BM * 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = shared ? (int)blockIdx.x % shared : (int)blockIdx.x, drow = lane >> 2, pc = lane & 3;   // shared: every workgroup reads one of `shared` tiles (the WEIGHT operand: L2 hits)
  const int row = tile * BM + wave * 16 + drow;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A), 0, bytes, 0x00020000);
  auto off = [&](int kt) -> unsigned {
    if (row >= M) return 0x80000000u;
    if (MODE == 0) return (unsigned)row * (unsigned)(nk * 64) + (unsigned)kt * 64u + (unsigned)pc * 16u;
    if (MODE == 1) return ((unsigned)(tile * nk + kt) * (unsigned)BM + (unsigned)(wave * 16 + drow)) * 64u + (unsigned)pc * 16u;
    return ((unsigned)(row / 32) * (unsigned)(nk / 2) + (unsigned)(kt / 2)) * 4096u + (unsigned)(row % 32) * 128u + (unsigned)(kt & 1) * 64u + (unsigned)pc * 16u;
  };
  const unsigned lds0 = (unsigned)(uintptr_t)LDSP(&st[0][0]) + (unsigned)wave * 1024u;
#pragma unroll
  for (int s = 0; s < NS - 1; s++) dma16(rs, lds0 + (unsigned)s * (BM * 64), off(s));
  unsigned acc = 0;
  for (int kt = 0; kt < nk; kt += NS) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int nx = kt + s + NS - 1;
      dma16(rs, lds0 + (unsigned)((s + NS - 1) % NS) * (BM * 64), off(nx < nk ? nx : nk - 1));
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NS - 1) : "memory");
      acc += *reinterpret_cast<const unsigned*>(&st[s][tid * 4 % (BM * 64)]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}

// writer: a wave owns 32 rows x 256 channels (8 patches of 32 x 32 channels = 32 rows x 128 B), four 16-byte stores per lane and patch
template <int MODE>
__global__ __launch_bounds__(256) void write_patches(unsigned char* O, int M, int ncb /* 32-channel blocks per row */, int cb0span) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prow = lane >> 3, chunk = lane & 7;
  const int g = (int)blockIdx.x * 4 + wave;                 // wave id -> (row block of 32, group of cb0span channel blocks)
  const int groups = ncb / cb0span, rb = g / groups, cg = g % groups;
  if (rb * 32 >= M) return;
  const uint4 v = make_uint4(0x39993999u + lane, 0x39993999u, 0x39993999u, 0x39993999u);
  for (int j = 0; j < cb0span; j++) {
    const int cb = cg * cb0span + j;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = rb * 32 + it * 8 + prow;
      size_t o;
      if (MODE == 0) o = (size_t)r * (size_t)(ncb * 128) + (size_t)cb * 128 + (size_t)chunk * 16;
      else o = ((size_t)rb * ncb + cb) * 4096 + (size_t)(it * 8 + prow) * 128 + (size_t)chunk * 16;
      *reinterpret_cast<uint4*>(O + o) = v;
    }
  }
}

template <typename F> static float best_of(F f, int reps = 7) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < reps; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
  return best;
}

int main() {
  const int M = 128000, K = 1024, nk = K / 16;
  const size_t bytes = (size_t)M * K * 4;
  unsigned char* A; unsigned* sink; hipMalloc(&A, bytes); hipMalloc(&sink, 64); hipMemset(A, 1, bytes);
  const double gb = bytes / 1e9;
  printf("pixel operand of a 1x1 convolution, M = %d pixels x K = %d channels (%.2f GB), k-tiles of 16 channels, LDS-DMA, no multiply\n", M, K, gb);
#define RD(MODE, NW, NS, NAME) { const float ms = best_of([&] { hipLaunchKernelGGL((read_tiles<MODE, NW, NS>), dim3((M + NW * 16 - 1) / (NW * 16)), dim3(NW * 64), 0, 0, A, M, nk, (unsigned)bytes, sink, SHARED); }); \
    printf("  read  %-44s %d-row tiles, %d stages%s: %.3f ms  %.0f GB/s\n", NAME, NW * 16, NS, SHARED ? " (every workgroup the SAME tile: weight operand, L2 hits)" : "", ms, gb / ms * 1e3); }
  int SHARED = 0;
  RD(0, 16, 4, "row-major NHWC (4 KB pixel pitch)")
  RD(1, 16, 4, "k-tile-major per tile (16 KB pieces)")
  RD(2, 16, 4, "32 x 32 blocks (4 KB pieces)")
  RD(0, 8, 3, "row-major NHWC (4 KB pixel pitch)")
  RD(1, 8, 3, "k-tile-major per tile (8 KB pieces)")
  RD(2, 8, 3, "32 x 32 blocks (4 KB pieces)")
  SHARED = 1;
  RD(0, 16, 4, "row-major [Cout][K] (4 KB row pitch)")
  RD(1, 16, 4, "k-tile-major (16 KB pieces)")
  SHARED = 4;
  RD(0, 16, 4, "row-major [Cout][K] (4 KB row pitch), 4 tiles")
  RD(1, 16, 4, "k-tile-major (16 KB pieces), 4 tiles")
  const int ncb = K / 32;
#define WR(MODE, SPAN, NAME) { const int waves = (M / 32) * (ncb / SPAN); const float ms = best_of([&] { hipLaunchKernelGGL((write_patches<MODE>), dim3((waves + 3) / 4), dim3(256), 0, 0, A, M, ncb, SPAN); }); \
    printf("  write %-44s a wave = 32 rows x %d channels: %.3f ms  %.0f GB/s\n", NAME, SPAN * 32, ms, gb / ms * 1e3); }
  WR(0, 2, "row-major NHWC (8 rows x 128 B per store)")
  WR(1, 2, "32 x 32 blocks (1 KB per store)")
  WR(0, 8, "row-major NHWC (8 rows x 128 B per store)")
  WR(1, 8, "32 x 32 blocks (1 KB per store)")
  return 0;
}
