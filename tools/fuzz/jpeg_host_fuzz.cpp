// libFuzzer + ASan/UBSan harness for the native JPEG marker walk (csrc/jpeg_host.hip): the parser reads untrusted files on the host,
// so every malformed input must end in "file_status = 1" (the file goes to Pillow), never in an out-of-bounds access.
// Build + run: tools/fuzz/run_jpeg_host_fuzz.sh [seconds].  Host code only, no GPU.
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../self-similarity-grouping_amd/csrc/jpeg_host.hip"

void ssg_set_error(const char*, ...) {}

static void one_batch(const std::vector<const void*>& files, const std::vector<int64_t>& lens, int nthreads) {
  void* h = nullptr;
  int64_t c[10];
  std::vector<int32_t> st(files.size());
  if (ssg_jpeg_parse_open(files.data(), lens.data(), (int)files.size(), nthreads, &h, c, st.data()) != SSG_OK) return;
  // exactly the sizes the header documents: an overrun of any of them is a finding
  std::vector<int64_t> imgs(c[0] * 32 + 1), segs(c[1] * 5 + 1);
  std::vector<uint8_t> pool(c[2] + 1), vals(c[3] * 256 + 1);
  std::vector<uint16_t> look(c[3] * 256 + 1), qts(c[4] * 64 + 1);
  std::vector<int32_t> maxcode(c[3] * 18 + 1), valoff(c[3] * 17 + 1);
  ssg_jpeg_parse_fill(h, imgs.data(), segs.data(), pool.data(), look.data(), maxcode.data(), valoff.data(), vals.data(), qts.data());
  // what the device kernel relies on: segments inside the pool, tables inside their arrays
  for (int64_t s = 0; s < c[1]; s++) {
    const int64_t* g = &segs[s * 5];
    if (g[0] < 0 || g[0] >= c[0] || g[1] < 0 || g[2] < 0 || g[1] + g[2] > c[2] - 64) __builtin_trap();
  }
  for (int64_t i = 0; i < c[0]; i++) {
    const int64_t* im = &imgs[i * 32];
    if (im[0] <= 0 || im[1] <= 0 || im[2] < 1 || im[2] > 3) __builtin_trap();
    for (int ci = 0; ci < im[2]; ci++) {
      const int64_t* cd = im + 8 + 8 * ci;
      if (cd[5] < 0 || cd[5] >= c[4] || cd[6] < 0 || cd[6] >= c[3] || cd[7] < 0 || cd[7] >= c[3]) __builtin_trap();
    }
  }
  ssg_jpeg_parse_close(h);
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size) {
  // exact-size heap copies so that ASan sees a read one byte past the file
  std::vector<uint8_t> a(data, data + size);
  const size_t cut = size / 2;
  std::vector<uint8_t> b(data, data + cut);
  one_batch({a.data()}, {(int64_t)a.size()}, 1);
  one_batch({a.data(), b.data(), nullptr}, {(int64_t)a.size(), (int64_t)b.size(), 0}, 2);
  return 0;
}
