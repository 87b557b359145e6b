// jpeg_host.hip -- host side of the JPEG input path: marker walk + table building for a BATCH of files, in native code on several threads.
//
// The reference decodes its images inside DataLoader workers (selftraining.py:49-53: `DataLoader(Preprocessor(...), num_workers=...)`,
// libjpeg behind Pillow) -- several processes of C.  Here the entropy decode runs on the GPU (jpeg.hip), so what is left on the host is the
// bookkeeping: walk the marker segments of every file, build the derived Huffman tables, find the restart segments and lay the batch
// out for ssg_jpeg_decode_batch.  ssg_amd/jpeg.py states that bookkeeping in Python (scan_header / parse_batch: ~55 us per Market-1501
// file, 17-20 k files/s on one core -- below the embedder's 26 k images/s); this file is the same walk in C++ on `nthreads` threads,
// checked against the Python statement field by field (tests/test_oracle_golden.py::test_jpeg_native_parser_matches_python).
// Host code only: no kernel, no device memory.
#include "ssg_common.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace ssg {
namespace jpegh {

constexpr int IMG_WORDS = 32, SEG_WORDS = 5;
static const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff { bool present = false; unsigned char counts[16]; unsigned char symbols[256]; int nsym = 0; };
struct Hdr {
  bool ok = false;
  int width = 0, height = 0, ncomp = 0, ri = 0;
  int comp[3][4];          // id, h, v, tq
  int scan[3][3];          // component index, dc table, ac table
  bool qt_present[4] = {false, false, false, false};
  uint16_t qt[4][64];
  Huff dc[4], ac[4];
  int64_t ecs_start = 0, ecs_end = 0;
  int64_t nseg = 0;        // restart segments of the scan
};

static inline int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

// ssg_amd/jpeg.py scan_header: returns false for anything the GPU decoder does not take (the file then goes to Pillow)
static bool scan_header(const unsigned char* b, int64_t n, Hdr& h) {
  if (n < 4 || b[0] != 0xFF || b[1] != 0xD8) return false;
  int adobe = -1; bool jfif = false, have_frame = false;
  int64_t p = 2;
  for (;;) {
    while (p < n && b[p] != 0xFF) p++;
    while (p < n && b[p] == 0xFF) p++;
    if (p >= n) return false;
    const int marker = b[p++];
    if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD8)) continue;
    if (marker == 0xD9 || p + 2 > n) return false;
    const int length = be16(b + p);
    if (length < 2 || p + length > n) return false;
    const unsigned char* body = b + p + 2;
    const int blen = length - 2;
    p += length;
    if (marker == 0xDB) {                                     // DQT
      int q = 0;
      while (q < blen) {
        const int prec = body[q] >> 4, tid = body[q] & 15;
        q++;
        if (tid > 3 || prec > 1) return false;
        const int need = prec ? 128 : 64;
        if (q + need > blen) return false;
        for (int i = 0; i < 64; i++) h.qt[tid][kZigzag[i]] = (uint16_t)(prec ? be16(body + q + 2 * i) : body[q + i]);
        h.qt_present[tid] = true;
        q += need;
      }
    } else if (marker == 0xC4) {                              // DHT
      int q = 0;
      while (q < blen) {
        const int cls = body[q] >> 4, tid = body[q] & 15;
        if (q + 17 > blen || cls > 1 || tid > 3) return false;
        const unsigned char* counts = body + q + 1;
        int total = 0;
        for (int i = 0; i < 16; i++) total += counts[i];
        if (total > 256 || q + 17 + total > blen) return false;
        const unsigned char* sym = body + q + 17;
        if (cls == 0) for (int i = 0; i < total; i++) if (sym[i] > 15) return false;
        int code = 0;
        for (int len = 1; len <= 16; len++) { code += counts[len - 1]; if (code > (1 << len)) return false; code <<= 1; }
        Huff& t = (cls ? h.ac : h.dc)[tid];
        t.present = true; memcpy(t.counts, counts, 16); memset(t.symbols, 0, 256); memcpy(t.symbols, sym, (size_t)total); t.nsym = total;
        q += 17 + total;
      }
    } else if (marker == 0xC0 || marker == 0xC1) {            // baseline / extended sequential, Huffman
      if (blen < 6 || body[0] != 8) return false;
      h.height = be16(body + 1); h.width = be16(body + 3);
      h.ncomp = body[5];
      if (blen < 6 + 3 * h.ncomp) return false;
      if (h.ncomp != 1 && h.ncomp != 3) return false;
      for (int i = 0; i < h.ncomp; i++) { h.comp[i][0] = body[6 + 3 * i]; h.comp[i][1] = body[7 + 3 * i] >> 4; h.comp[i][2] = body[7 + 3 * i] & 15; h.comp[i][3] = body[8 + 3 * i]; }
      have_frame = true;
    } else if (marker >= 0xC2 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
      return false;
    } else if (marker == 0xDD) {
      if (blen < 2) return false;
      h.ri = be16(body);
    } else if (marker == 0xEE && blen >= 12 && memcmp(body, "Adobe", 5) == 0) {
      adobe = body[11];
    } else if (marker == 0xE0 && blen >= 5 && memcmp(body, "JFIF\0", 5) == 0) {
      jfif = true;
    } else if (marker == 0xDA) {                              // SOS: must be the single scan of a sequential file
      if (!have_frame || blen < 1) return false;
      const int ns = body[0];
      if (ns != h.ncomp || blen < 3 + 2 * ns) return false;
      for (int i = 0; i < ns; i++) {
        int idx = -1;
        for (int c = 0; c < h.ncomp; c++) if (h.comp[c][0] == body[1 + 2 * i]) { idx = c; break; }
        if (idx != i) return false;                           // unknown component, or not in frame order
        h.scan[i][0] = idx; h.scan[i][1] = body[2 + 2 * i] >> 4; h.scan[i][2] = body[2 + 2 * i] & 15;
      }
      if (body[1 + 2 * ns] != 0 || body[2 + 2 * ns] != 63) return false;
      h.ecs_start = p;
      break;
    }
  }
  if (h.ncomp == 3) {
    if (!jfif && adobe < 0 && h.comp[0][0] == 82 && h.comp[1][0] == 71 && h.comp[2][0] == 66) return false;   // RGB component ids
    if (!(adobe < 0 || adobe == 1)) return false;
    if (h.comp[1][1] != 1 || h.comp[1][2] != 1 || h.comp[2][1] != 1 || h.comp[2][2] != 1) return false;
    const int hs = h.comp[0][1], vs = h.comp[0][2];
    if (!((hs == 1 && vs == 1) || (hs == 2 && vs == 1) || (hs == 2 && vs == 2))) return false;
  } else {
    h.comp[0][1] = 1; h.comp[0][2] = 1;
  }
  if (h.width == 0 || h.height == 0) return false;
  for (int i = 0; i < h.ncomp; i++) {
    const int tq = h.comp[i][3];
    if (tq > 3 || !h.qt_present[tq]) return false;
    if (h.scan[i][1] > 3 || h.scan[i][2] > 3 || !h.dc[h.scan[i][1]].present || !h.ac[h.scan[i][2]].present) return false;
  }
  // end of the entropy-coded data: the first marker that is neither a stuffed zero nor RSTn
  const unsigned char* e = b + h.ecs_start;
  const int64_t len = n - h.ecs_start;
  int64_t q = 0;
  for (;;) {
    const unsigned char* f = q < len ? (const unsigned char*)memchr(e + q, 0xFF, (size_t)(len - q)) : nullptr;
    if (!f || (f - e) + 1 >= len) { q = len; break; }
    q = f - e;
    if (e[q + 1] == 0 || (e[q + 1] >= 0xD0 && e[q + 1] <= 0xD7)) { q += 2; continue; }
    break;
  }
  h.ecs_end = h.ecs_start + q;
  h.ok = true;
  return true;
}

// restart segments of one scan, exactly like parse_batch: visit(start, length, first_mcu, mcu_count)
template <class F>
static void walk_segments(const unsigned char* ecs, int64_t len, int ri_hdr, int nmcu, F visit) {
  const int ri = ri_hdr ? ri_hdr : nmcu;
  int64_t start = 0; int m0 = 0;
  for (int64_t i = 0; i + 1 < len;) {
    if (ecs[i] == 0xFF && ecs[i + 1] >= 0xD0 && ecs[i + 1] <= 0xD7) {
      visit(start, i - start, m0, std::min(ri, nmcu - m0));
      start = i + 2; m0 += ri; i += 2;
      if (m0 >= nmcu) break;
    } else {
      const unsigned char* f = (const unsigned char*)memchr(ecs + i + (ecs[i] == 0xFF ? 1 : 0), 0xFF, (size_t)(len - i - (ecs[i] == 0xFF ? 1 : 0)));
      if (!f) break;
      i = f - ecs;
    }
  }
  if (m0 < nmcu) visit(start, len - start, m0, std::min(ri, nmcu - m0));
}

struct Batch {
  int nfiles = 0;
  std::vector<const unsigned char*> data; std::vector<int64_t> len;
  std::vector<Hdr> hdr;
  std::vector<int> kept;                              // file index of image k
  std::vector<int64_t> ecs_off, seg_off;              // per image
  std::vector<std::string> huff_keys, qt_keys;        // de-duplicated tables in first-use order
  std::vector<int64_t> imgs;                          // [nkept][32]
  int64_t counts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int nthreads = 1;
};

template <class F>
static void parallel_for(int n, int nthreads, F body) {
  nthreads = std::max(1, std::min(nthreads, n));
  if (nthreads == 1) { for (int i = 0; i < n; i++) body(i); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([=]() { for (int i = (int)((int64_t)n * t / nthreads); i < (int)((int64_t)n * (t + 1) / nthreads); i++) body(i); });
  for (auto& x : th) x.join();
}

// jdhuff.c jpeg_make_d_derived_tbl (ssg_amd/jpeg.py _derived)
static void derive(const unsigned char* counts, const unsigned char* symbols, uint16_t* look, int32_t* maxcode, int32_t* valoff, uint8_t* vals) {
  memset(look, 0, 256 * sizeof(uint16_t)); memset(valoff, 0, 17 * sizeof(int32_t));
  for (int i = 0; i < 18; i++) maxcode[i] = -1;
  memcpy(vals, symbols, 256);
  int code = 0, p = 0;
  for (int length = 1; length <= 16; length++) {
    const int cnt = counts[length - 1];
    if (cnt) {
      valoff[length] = p - code;
      if (length <= 8)
        for (int i = 0; i < cnt; i++) {
          const int first = (code + i) << (8 - length);
          for (int j = 0; j < (1 << (8 - length)); j++) look[first + j] = (uint16_t)((length << 8) | symbols[p + i]);
        }
      code += cnt; p += cnt;
      maxcode[length] = code - 1;
    }
    code <<= 1;
  }
  maxcode[17] = 0xFFFFF;
}

}  // namespace jpegh
}  // namespace ssg

using namespace ssg;

// Parse a batch of files held in host memory.  files[i] / lens[i]: the i-th file's bytes (they must stay valid until ssg_jpeg_parse_close).
// Pass 1 (nthreads threads): marker walk of every file; pass 2: batch layout (block / plane / output offsets, de-duplicated tables).
// counts10 (out): images the GPU decodes, restart segments, bytes of entropy-coded data (incl. 64 bytes of padding), Huffman tables,
//   quantisation tables, coefficient blocks, largest component in blocks, plane bytes, output bytes, largest image in pixels
//   -- the sizes of the buffers ssg_jpeg_parse_fill writes and ssg_jpeg_decode_batch takes.
// file_status (out, int32 [nfiles]): 0 = decoded on the GPU (it is image number `rank among the zeros`), 1 = left to the reference's decoder.
extern "C" int ssg_jpeg_parse_open(const void* const* files, const int64_t* lens, int nfiles, int nthreads, void** handle, int64_t* counts10,
                                   int32_t* file_status) {
  using namespace jpegh;
  if (!files || !lens || nfiles <= 0 || !handle || !counts10 || !file_status) { ssg_set_error("ssg_jpeg_parse_open: bad arguments"); return SSG_ERR_INVALID; }
  Batch* B = new Batch();
  B->nfiles = nfiles; B->nthreads = nthreads > 0 ? nthreads : 1;
  B->data.resize(nfiles); B->len.assign(lens, lens + nfiles); B->hdr.resize(nfiles);
  for (int i = 0; i < nfiles; i++) B->data[i] = (const unsigned char*)files[i];
  parallel_for(nfiles, B->nthreads, [B](int i) {
    Hdr& h = B->hdr[i];
    if (!B->data[i] || B->len[i] <= 0 || !scan_header(B->data[i], B->len[i], h)) { h.ok = false; return; }
    const int hs = h.comp[0][1], vs = h.comp[0][2];
    const int mcux = (h.width + 8 * hs - 1) / (8 * hs), mcuy = (h.height + 8 * vs - 1) / (8 * vs);
    int64_t ns = 0;
    walk_segments(B->data[i] + h.ecs_start, h.ecs_end - h.ecs_start, h.ri, mcux * mcuy, [&](int64_t, int64_t, int, int) { ns++; });
    h.nseg = ns;
  });
  std::map<std::string, int> huff_ids, qt_ids;
  int64_t ecs_off = 0, blocks = 0, plane_off = 0, out_off = 0, max_blocks = 0, max_pixels = 0, nseg = 0;
  for (int i = 0; i < nfiles; i++) {
    const Hdr& h = B->hdr[i];
    file_status[i] = h.ok ? 0 : 1;
    if (!h.ok) continue;
    B->kept.push_back(i);
    const int hs = h.comp[0][1], vs = h.comp[0][2];
    const int mcux = (h.width + 8 * hs - 1) / (8 * hs), mcuy = (h.height + 8 * vs - 1) / (8 * vs);
    int64_t im[IMG_WORDS];
    memset(im, 0, sizeof(im));
    im[0] = h.width; im[1] = h.height; im[2] = h.ncomp; im[3] = hs; im[4] = vs; im[5] = mcux; im[6] = mcuy; im[7] = out_off;
    for (int ci = 0; ci < h.ncomp; ci++) {
      const int bw = mcux * h.comp[ci][1], bh = mcuy * h.comp[ci][2];
      const std::string qk((const char*)h.qt[h.comp[ci][3]], 128);
      auto qi = qt_ids.find(qk);
      if (qi == qt_ids.end()) { qi = qt_ids.emplace(qk, (int)B->qt_keys.size()).first; B->qt_keys.push_back(qk); }
      int hid[2];
      for (int w = 0; w < 2; w++) {
        const Huff& t = w ? h.ac[h.scan[ci][2]] : h.dc[h.scan[ci][1]];
        std::string hk((const char*)t.counts, 16); hk.append((const char*)t.symbols, 256);
        auto hi = huff_ids.find(hk);
        if (hi == huff_ids.end()) { hi = huff_ids.emplace(hk, (int)B->huff_keys.size()).first; B->huff_keys.push_back(hk); }
        hid[w] = hi->second;
      }
      int64_t* cd = im + 8 + 8 * ci;
      cd[0] = blocks; cd[1] = bw; cd[2] = bh; cd[3] = plane_off; cd[4] = bw * 8; cd[5] = qi->second; cd[6] = hid[0]; cd[7] = hid[1];
      blocks += (int64_t)bw * bh; plane_off += (int64_t)bw * bh * 64;
      max_blocks = std::max<int64_t>(max_blocks, (int64_t)bw * bh);
    }
    out_off += (int64_t)h.width * h.height * 3;
    max_pixels = std::max<int64_t>(max_pixels, (int64_t)h.width * h.height);
    B->imgs.insert(B->imgs.end(), im, im + IMG_WORDS);
    B->ecs_off.push_back(ecs_off); B->seg_off.push_back(nseg);
    ecs_off += h.ecs_end - h.ecs_start; nseg += h.nseg;
  }
  int64_t* c = B->counts;
  c[0] = (int64_t)B->kept.size(); c[1] = nseg; c[2] = ecs_off + 64; c[3] = (int64_t)B->huff_keys.size(); c[4] = (int64_t)B->qt_keys.size();
  c[5] = blocks; c[6] = max_blocks; c[7] = plane_off; c[8] = out_off; c[9] = max_pixels;
  memcpy(counts10, c, sizeof(B->counts));
  *handle = B;
  return SSG_OK;
}

// Pass 3 (nthreads threads): write the batch description into the caller's host buffers, sized by counts10 of ssg_jpeg_parse_open:
//   imgs int64 [c0][32], segs int64 [c1][5], pool uint8 [c2] (entropy-coded bytes back to back + zero padding), look uint16 [c3][256],
//   maxcode int32 [c3][18], valoff int32 [c3][17], vals uint8 [c3][256], qts uint16 [c4][64] -- the arguments of ssg_jpeg_decode_batch.
extern "C" int ssg_jpeg_parse_fill(void* handle, int64_t* imgs, int64_t* segs, uint8_t* pool, uint16_t* look, int32_t* maxcode, int32_t* valoff,
                                   uint8_t* vals, uint16_t* qts) {
  using namespace jpegh;
  Batch* B = (Batch*)handle;
  if (!B) { ssg_set_error("ssg_jpeg_parse_fill: null handle"); return SSG_ERR_INVALID; }
  const int nk = (int)B->kept.size();
  if (nk == 0) return SSG_OK;
  if (!imgs || !segs || !pool || !look || !maxcode || !valoff || !vals || !qts) { ssg_set_error("ssg_jpeg_parse_fill: null buffer"); return SSG_ERR_INVALID; }
  memcpy(imgs, B->imgs.data(), B->imgs.size() * sizeof(int64_t));
  for (size_t t = 0; t < B->huff_keys.size(); t++) {
    const unsigned char* k = (const unsigned char*)B->huff_keys[t].data();
    derive(k, k + 16, look + t * 256, maxcode + t * 18, valoff + t * 17, vals + t * 256);
  }
  for (size_t t = 0; t < B->qt_keys.size(); t++) memcpy(qts + t * 64, B->qt_keys[t].data(), 128);
  memset(pool + B->counts[2] - 64, 0, 64);
  parallel_for(nk, B->nthreads, [=](int k) {
    const int i = B->kept[k];
    const Hdr& h = B->hdr[i];
    const unsigned char* ecs = B->data[i] + h.ecs_start;
    const int64_t len = h.ecs_end - h.ecs_start, off = B->ecs_off[k];
    memcpy(pool + off, ecs, (size_t)len);
    const int hs = h.comp[0][1], vs = h.comp[0][2];
    const int mcux = (h.width + 8 * hs - 1) / (8 * hs), mcuy = (h.height + 8 * vs - 1) / (8 * vs);
    int64_t* s = segs + B->seg_off[k] * SEG_WORDS;
    walk_segments(ecs, len, h.ri, mcux * mcuy, [&](int64_t start, int64_t slen, int m0, int cnt) {
      s[0] = k; s[1] = off + start; s[2] = slen; s[3] = m0; s[4] = cnt; s += SEG_WORDS;
    });
  });
  return SSG_OK;
}

extern "C" int ssg_jpeg_parse_close(void* handle) {
  delete (ssg::jpegh::Batch*)handle;
  return SSG_OK;
}
