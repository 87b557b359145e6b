// libFuzzer + ASan harness for the DEVICE side of the JPEG path, run on the host: the three kernels of csrc/jpeg.hip use no wave-level
// operation (one thread per restart segment / 8x8 block / pixel), so their source text (cut out of jpeg.hip by run_jpeg_device_fuzz.sh
// into jpeg_kernels_cut.inc, `__global__` / `__device__` mapped to plain host functions) can be executed thread by thread on the CPU
// under the sanitizers.  Input: a (possibly damaged) file -> the native parser (csrc/jpeg_host.hip) -> exactly-sized buffers -> every
// thread of the three launches of ssg_jpeg_decode_batch.  A finding = an access outside the buffers that entry point documents (pool
// incl. its padding, coefficients, planes, tables, pixels), undefined arithmetic, or a segment that does not terminate.
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../self-similarity-grouping_amd/csrc/jpeg_host.hip"     // (brings hip_runtime.h for the host)
void ssg_set_error(const char*, ...) {}

struct FzDim { unsigned x = 0, y = 0, z = 0; };
static FzDim fz_blockIdx, fz_blockDim, fz_threadIdx;
static inline int fz_atomicOr(int32_t* p, int v) { const int o = *p; *p |= v; return o; }
#undef __device__
#undef __global__
#undef __constant__
#undef __forceinline__
#define __device__
#define __global__
#define __constant__ static const
#define __forceinline__ inline
#undef __launch_bounds__
#define __launch_bounds__(x)
#define blockIdx fz_blockIdx
#define blockDim fz_blockDim
#define threadIdx fz_threadIdx
#define atomicOr fz_atomicOr
#include "jpeg_kernels_cut.inc"

static std::vector<uint8_t>* g_pixels = nullptr;   // FZ_MAIN: where the decoded RGB bytes go
static int g_status = -1;

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size) {
  std::vector<uint8_t> a(data, data + size);
  const void* files[1] = {a.data()};
  const int64_t lens[1] = {(int64_t)a.size()};
  void* h = nullptr;
  int64_t c[10];
  int32_t st[1];
  if (ssg_jpeg_parse_open(files, lens, 1, 1, &h, c, st) != SSG_OK) return 0;
  if (c[0] == 1 && c[5] <= 4096 && c[9] <= (1 << 18)) {   // (bounded: a header may announce 65535 x 65535 pixels)
    std::vector<int64_t> imgs(c[0] * 32), segs(c[1] * 5);
    std::vector<uint8_t> pool(c[2]), vals(c[3] * 256);
    std::vector<uint16_t> look(c[3] * 256), qts(c[4] * 64);
    std::vector<int32_t> maxcode(c[3] * 18), valoff(c[3] * 17), status(c[0], 0);
    std::vector<int16_t> coef(c[5] * 64, 0);
    std::vector<uint8_t> planes(c[7]), out(c[8]);
    ssg_jpeg_parse_fill(h, imgs.data(), segs.data(), pool.data(), look.data(), maxcode.data(), valoff.data(), vals.data(), qts.data());
    ssg::jpeg::Tables t{look.data(), maxcode.data(), valoff.data(), vals.data()};
    fz_blockDim.x = 1;
    for (int64_t s = 0; s < c[1]; s++) {
      fz_blockIdx.x = (unsigned)s;
      ssg::jpeg::huffman_kernel(pool.data(), segs.data(), (int)c[1], imgs.data(), t, coef.data(), status.data());
    }
    for (unsigned y = 0; y < 3; y++)                      // grid (ceil(max_blocks / 64), nimg * 3) x 64 threads
      for (int64_t b = 0; b < (c[6] + 63) / 64 * 64; b++) {
        fz_blockIdx.x = (unsigned)b; fz_blockIdx.y = y;
        ssg::jpeg::idct_kernel(imgs.data(), coef.data(), qts.data(), planes.data());
      }
    fz_blockIdx.y = 0;
    for (int64_t px = 0; px < (c[9] + 255) / 256 * 256; px++) {   // grid (ceil(max_pixels / 256), nimg) x 256 threads
      fz_blockIdx.x = (unsigned)px;
      ssg::jpeg::colour_kernel(imgs.data(), planes.data(), out.data());
    }
    if (g_pixels) { *g_pixels = out; g_status = status[0]; }
  }
  ssg_jpeg_parse_close(h);
  return 0;
}

#ifdef FZ_MAIN
// jpeg_device_host <in.jpg> <out.rgb>: the kernels' source text as a host decoder (tests/test_oracle_golden.py compares it with Pillow).
// exit 0 = decoded, 3 = the parser hands the file to Pillow, 4 = the Huffman kernel flagged damaged data
int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  std::vector<uint8_t> px;
  g_pixels = &px;
  LLVMFuzzerTestOneInput(buf.data(), buf.size());
  if (g_status < 0) return 3;
  if (g_status) return 4;
  f = fopen(argv[2], "wb");
  fwrite(px.data(), 1, px.size(), f);
  fclose(f);
  return 0;
}
#endif
