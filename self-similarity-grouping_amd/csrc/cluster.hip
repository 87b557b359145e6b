// cluster.hip -- K10 (epsilon rule), K11 (DBSCAN region query), K12 (connected components).
//
// K10 replaces selftraining.py:289-293:
//   tri = triu(dist, 1); tri = tri[nonzero(tri)]; tri = sort(tri)
//   eps = tri[:round(rho * tri.size)].mean()
// as a 12-bit-digit radix select over the monotone u64 patterns of the (positive) float64
// distances, a compaction of everything at or below the selected digit, a bitonic sort of
// that small set and numpy's pairwise-summation mean -- bit-identical to np.mean.
//
// K11/K12 replace sklearn 1.7.2 DBSCAN(eps, min_samples, metric='precomputed').fit_predict
// (selftraining.py:295,306): neighbourhood = {k : d[i,k] <= eps} row-wise, core = count >=
// min_samples, clusters = connected components of core-core edges numbered by their
// smallest core index (== dbscan_inner's visiting order), border points take the smallest
// cluster id among their core neighbours, everything else is -1.
//
// The matrix is never materialised in f64: mode 0 rebuilds final_dist from the compact
// half J' and the source vector v (reid/rerank.py:122), mode 1 reads a plain half matrix
// (the no-rerank euclidean_dist).  All matrix passes are HBM-read-bound (2 bytes/entry).
#include "ssg_common.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace ssg {

struct MatView {
  const void* M;       // [nrows, N] rows of this row block: half (mode 0/1) or double (mode 2)
  const hbits* v;      // [N] source vector (mode 0) or null
  int N, row0, nrows, mode;
  double lambda_value;
};

// Row streaming shared by the matrix passes: one wave per row, 8 consecutive elements per
// lane per chunk (16 B of half / 64 B of double), chunks aligned to 8 elements of the block.
struct RowStream {
  int64_t total, al; int first, nchunks;
  __device__ __forceinline__ RowStream(int row, int N, int nrows) {
    total = (int64_t)nrows * N;
    const int64_t base = (int64_t)row * N;
    al = base & ~(int64_t)7; first = (int)(base - al); nchunks = (first + N + 511) / 512;
  }
  __device__ __forceinline__ int col0(int c, int lane) const { return c * 512 + lane * 8 - first; }
};

// d[e] = value of column col0+e of row gi (garbage for columns outside [0,N): callers mask)
template <int MODE>
__device__ __forceinline__ void load_vals(const MatView& mv, const RowStream& rs, int c, int lane, int gi, double d[8]) {
  const int64_t off = rs.al + (int64_t)c * 512 + lane * 8;
  if (MODE == 2) {
    const double* M = reinterpret_cast<const double*>(mv.M);
    if (off + 8 <= rs.total) {
#pragma unroll
      for (int q = 0; q < 4; q++) { const double2 t = *reinterpret_cast<const double2*>(M + off + 2 * q); d[2 * q] = t.x; d[2 * q + 1] = t.y; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) d[e] = (off + e < rs.total) ? M[off + e] : 0.0;
    }
    return;
  }
  const hbits* M = reinterpret_cast<const hbits*>(mv.M);
  unsigned w[4] = {0, 0, 0, 0};
  if (off + 8 <= rs.total) { const uint4 x = *reinterpret_cast<const uint4*>(M + off); w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; }
  else if (off < rs.total) {
#pragma unroll
    for (int e = 0; e < 8; e++) if (off + e < rs.total) w[e >> 1] |= (unsigned)M[off + e] << ((e & 1) * 16);
  }
  const int j0 = rs.col0(c, lane);
  unsigned vw[4] = {0, 0, 0, 0};   // v[j0 .. j0+7] (mode 0)
  if (MODE == 0) {
    if (j0 >= 0 && j0 + 8 <= mv.N && (j0 & 7) == 0) {   // rows start on 16-byte boundaries when N % 8 == 0: one load
      const uint4 x = *reinterpret_cast<const uint4*>(mv.v + j0); vw[0] = x.x; vw[1] = x.y; vw[2] = x.z; vw[3] = x.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) { const int k = j0 + e; if (k >= 0 && k < mv.N) vw[e >> 1] |= (unsigned)mv.v[k] << ((e & 1) * 16); }
    }
  }
  const hbits vi = MODE == 0 ? mv.v[gi] : (hbits)0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const hbits raw = (hbits)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    if (MODE == 0) d[e] = final_dist_value(raw, vi, (hbits)((vw[e >> 1] >> ((e & 1) * 16)) & 0xffffu), mv.lambda_value);
    else d[e] = (double)h2f(raw);
  }
}

// The same in two steps for the half modes, so that a streaming loop can keep several chunks' loads in flight before it
// decodes the first one (one 1 KiB load per wave at a time leaves the pass latency-bound at ~2.3 TB/s).
struct RawChunk { uint4 m, v; };
template <int MODE>
__device__ __forceinline__ RawChunk load_raw(const MatView& mv, const RowStream& rs, int c, int lane) {
  static_assert(MODE == 0 || MODE == 1, "half matrices only");
  RawChunk r; r.m = make_uint4(0, 0, 0, 0); r.v = make_uint4(0, 0, 0, 0);
  const int64_t off = rs.al + (int64_t)c * 512 + lane * 8;
  const hbits* M = reinterpret_cast<const hbits*>(mv.M);
  if (c < rs.nchunks) {
    if (off + 8 <= rs.total) r.m = *reinterpret_cast<const uint4*>(M + off);
    else if (off < rs.total) {
      unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 8; e++) if (off + e < rs.total) w[e >> 1] |= (unsigned)M[off + e] << ((e & 1) * 16);
      r.m = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (MODE == 0) {
      const int j0 = rs.col0(c, lane);
      if (j0 >= 0 && j0 + 8 <= mv.N && (j0 & 7) == 0) r.v = *reinterpret_cast<const uint4*>(mv.v + j0);
      else {
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; e++) { const int k = j0 + e; if (k >= 0 && k < mv.N) w[e >> 1] |= (unsigned)mv.v[k] << ((e & 1) * 16); }
        r.v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }
  return r;
}
template <int MODE>
__device__ __forceinline__ void decode_raw(const MatView& mv, const RawChunk& r, hbits vi, double d[8]) {
  const unsigned w[4] = {r.m.x, r.m.y, r.m.z, r.m.w}, vw[4] = {r.v.x, r.v.y, r.v.z, r.v.w};
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const hbits raw = (hbits)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    if (MODE == 0) d[e] = final_dist_value(raw, vi, (hbits)((vw[e >> 1] >> ((e & 1) * 16)) & 0xffffu), mv.lambda_value);
    else d[e] = (double)h2f(raw);
  }
}
constexpr int PIPE = 4;     // chunks (4 KiB of the matrix per wave) in flight in the streaming passes

// ------------------------------------------------------------------ K10 histogram level
// hist[bin] += #{ (i<k), d != 0, (key >> (shift+width)) == prefix }, bin = (key>>shift) & mask.
// hist[4096] += number of non-zero strict-upper entries (only when count_nonzero != 0).
template <int MODE>
__global__ __launch_bounds__(256) void eps_hist_kernel(MatView mv, unsigned long long prefix, int shift, int width, int count_nonzero,
                                                       unsigned long long* __restrict__ hist) {
  __shared__ unsigned int lh[4096];
  __shared__ unsigned long long lnz;
  for (int b = (int)threadIdx.x; b < 4096; b += 256) lh[b] = 0;
  if (threadIdx.x == 0) lnz = 0;
  __syncthreads();
  const int lane = lane_id();
  const unsigned mask = (1u << width) - 1u;
  const bool anyprefix = (shift + width) >= 63;   // top level: no prefix to match
  unsigned long long nz = 0;
  for (int il = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); il < mv.nrows; il += (int)gridDim.x * 4) {
    const int gi = mv.row0 + il;
    RowStream rs(il, mv.N, mv.nrows);
    // strict upper triangle: columns k > gi only -> skip leading chunks
    int c0 = (rs.first + gi + 1) / 512;
    const hbits vi = MODE == 0 ? mv.v[gi] : (hbits)0;
    for (int cb = c0; cb < rs.nchunks; cb += PIPE) {
      RawChunk raw[PIPE];
      if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < PIPE; u++) raw[u] = load_raw<MODE == 2 ? 1 : MODE>(mv, rs, cb + u, lane);
      }
#pragma unroll
      for (int u = 0; u < PIPE; u++) {
        const int c = cb + u;
        if (c >= rs.nchunks) break;
        double dv[8];
        if (MODE == 2) load_vals<MODE>(mv, rs, c, lane, gi, dv);
        else decode_raw<MODE == 2 ? 1 : MODE>(mv, raw[u], vi, dv);
        const int j0 = rs.col0(c, lane);
        int pbin = -1; unsigned pcnt = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int k = j0 + e;
          if (k > gi && k < mv.N) {
            const double d = dv[e];
            if (d != 0.0) {
              nz++;
              const unsigned long long key = (unsigned long long)__double_as_longlong(d);
              if (anyprefix || (key >> (shift + width)) == prefix) {
                const int bin = (int)((key >> shift) & mask);
                if (bin == pbin) pcnt++;
                else { if (pcnt) atomicAdd(&lh[pbin], pcnt); pbin = bin; pcnt = 1; }
              }
            }
          }
        }
        // the bulk of a row falls into one bin: one add per wave instead of 64
        const int fb = __shfl(pbin, 0, 64);
        if (__all(pbin == fb)) {
          unsigned tot = pcnt;
          for (int sh = 1; sh < 64; sh <<= 1) tot += (unsigned)__shfl_xor((int)tot, sh, 64);
          if (lane == 0 && fb >= 0 && tot) atomicAdd(&lh[fb], tot);
        } else if (pcnt) atomicAdd(&lh[pbin], pcnt);
      }
    }
  }
  for (int sh = 1; sh < 64; sh <<= 1) nz += (unsigned long long)__shfl_xor((long long)nz, sh, 64);
  if (lane == 0 && nz) atomicAdd(&lnz, nz);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < 4096; b += 256) if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
  if (threadIdx.x == 0 && count_nonzero && lnz) atomicAdd(&hist[4096], lnz);
}

// Per-wave staging buffer in LDS: results are appended with ballot prefix sums and flushed to a
// global list with ONE cursor atomic per ~512 entries (a per-chunk atomic on one word saturates
// at ~90 ops/us and was the bottleneck of the first version).
constexpr int STAGE_CAP = 1024;
template <typename T>
struct WaveStage {
  T* buf; int n;
  __device__ __forceinline__ void flush(T* __restrict__ out, unsigned long long cap, unsigned long long* __restrict__ cursor, int lane) {
    if (n == 0) return;
    unsigned long long basep = 0;
    if (lane == 0) basep = atomicAdd(cursor, (unsigned long long)n);
    basep = (unsigned long long)__shfl((long long)basep, 0, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int x = lane; x < n; x += 64) if (basep + x < cap) out[basep + x] = buf[x];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    n = 0;
  }
};

// Final flush of a workgroup's four wave stages with ONE cursor atomic (the sparse passes end with a few dozen staged entries per wave:
// a returning atomic per wave on one word -- ~90 per us -- was their whole run time, 5120 waves = 57 us).  Every thread of the block calls it.
template <typename T>
__device__ __forceinline__ void block_flush(WaveStage<T>& st, T* __restrict__ out, unsigned long long cap, unsigned long long* __restrict__ cursor) {
  __shared__ unsigned long long bf_base;
  __shared__ int bf_n[4];
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  if (lane == 0) bf_n[wave] = st.n;
  __syncthreads();
  if (threadIdx.x == 0) { const int tot = bf_n[0] + bf_n[1] + bf_n[2] + bf_n[3]; bf_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull; }
  __syncthreads();
  unsigned long long basep = bf_base;
  for (int w = 0; w < wave; w++) basep += (unsigned long long)bf_n[w];
  for (int x = lane; x < st.n; x += 64) if (basep + x < cap) out[basep + x] = st.buf[x];
  st.n = 0;
}

// append every strict-upper non-zero key <= key_max to buf
template <int MODE>
__global__ __launch_bounds__(256) void eps_compact_kernel(MatView mv, unsigned long long key_max, unsigned long long* __restrict__ buf,
                                                          unsigned long long cap, unsigned long long* __restrict__ cursor) {
  __shared__ unsigned long long sbuf[4][STAGE_CAP];
  const int lane = lane_id();
  WaveStage<unsigned long long> st{sbuf[threadIdx.x >> 6], 0};
  for (int il = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); il < mv.nrows; il += (int)gridDim.x * 4) {
    const int gi = mv.row0 + il;
    RowStream rs(il, mv.N, mv.nrows);
    int c0 = (rs.first + gi + 1) / 512;
    const hbits vi = MODE == 0 ? mv.v[gi] : (hbits)0;
    for (int cb = c0; cb < rs.nchunks; cb += PIPE) {
      RawChunk raw[PIPE];
      if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < PIPE; u++) raw[u] = load_raw<MODE == 2 ? 1 : MODE>(mv, rs, cb + u, lane);
      }
#pragma unroll
      for (int u = 0; u < PIPE; u++) {
        const int c = cb + u;
        if (c >= rs.nchunks) break;
        double dv[8];
        if (MODE == 2) load_vals<MODE>(mv, rs, c, lane, gi, dv);
        else decode_raw<MODE == 2 ? 1 : MODE>(mv, raw[u], vi, dv);
        const int j0 = rs.col0(c, lane);
        unsigned long long keys[8]; int n = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int k = j0 + e;
          keys[e] = ~0ULL;
          if (k > gi && k < mv.N) {
            const double d = dv[e];
            const unsigned long long key = (unsigned long long)__double_as_longlong(d);
            if (d != 0.0 && key <= key_max) { keys[e] = key; n++; }
          }
        }
        if (!__any(n > 0)) continue;
        int incl = n;
        for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_up(incl, sh, 64); if (lane >= sh) incl += o; }
        const int tot = __shfl(incl, 63, 64);
        int w = st.n + incl - n;
#pragma unroll
        for (int e = 0; e < 8; e++) if (keys[e] != ~0ULL) st.buf[w++] = keys[e];
        st.n += tot;
        if (st.n > STAGE_CAP - 512) st.flush(buf, cap, cursor, lane);
      }
    }
  }
  block_flush(st, buf, cap, cursor);
}

// ------------------------------------------------------------------ K10, fast path: sampled threshold + ONE full pass
// The radix select above costs two or three N^2 passes that each rebuild the float64 value of every element.  The fast path
// estimates the rho-quantile from a strided sample of rows (a few MB), turns it into a float32 threshold with a safety margin,
// and makes ONE pass over the strict upper triangle that (a) counts the exact zeros and (b) collects the exact float64 keys of
// the elements whose float32 SURROGATE value is below the threshold.  The host then sorts the collected keys and checks that
// the top-th smallest exact key lies below the threshold by more than the surrogate's error bound: in that case no
// uncollected element can be among the `top` smallest and the result is exactly the reference's; otherwise (a sample that
// underestimated the quantile) it falls back to the radix select.  Correctness never depends on the sample, only speed does.
//
// surrogate: float32 evaluation of the element (mode 0: jp + half(v_i+v_k)*lambda with float32 ops; modes 1/2: the value
// rounded to float32): |surrogate - exact| <= 4e-7 * (1 + |exact|) for the values that occur (final_dist < 4).
__device__ __forceinline__ int sur_bin(float x) {
  // 4096 bins over the float32 bit pattern: 16 binades [2^-14, 4) x 256 mantissa steps (0.4 % per bin); monotone in x >= 0
  const int b = (int)(__float_as_uint(x) >> 15) - 0x7100;
  return b < 0 ? 0 : (b > 4095 ? 4095 : b);
}
__device__ __forceinline__ float sur_bin_upper(int b) { return __uint_as_float((unsigned)(b + 1 + 0x7100) << 15); }   // exclusive upper edge of bin b

// raw half chunk -> 8 surrogate values (+ exact-zero mask for mode 0/1)
template <int MODE>
__device__ __forceinline__ void surrogate8(const MatView& mv, const RawChunk& r, hbits vi, float lam32, float sv[8], unsigned& zeromask) {
  const unsigned w[4] = {r.m.x, r.m.y, r.m.z, r.m.w}, vw[4] = {r.v.x, r.v.y, r.v.z, r.v.w};
  zeromask = 0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const hbits raw = (hbits)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    if (MODE == 0) {
      const hbits s = h_add((hbits)((vw[e >> 1] >> ((e & 1) * 16)) & 0xffffu), vi);
      const float p = h2f(s) * lam32;
      sv[e] = h2f(raw) + p;
      // exact value 0 <=> J' == +-0 and half(v_i+v_k)*lambda == 0 (all terms are finite and the sum of two non-zero terms of
      // opposite sign is decided exactly below, on the rare candidates only)
      if ((raw & 0x7fffu) == 0 && ((s & 0x7fffu) == 0 || mv.lambda_value == 0.0)) zeromask |= 1u << e;
    } else {
      sv[e] = h2f(raw);
      if ((raw & 0x7fffu) == 0) zeromask |= 1u << e;
    }
  }
}

// sample histogram: local rows 0, stride, 2*stride, ...; strict upper triangle, exact zeros dropped; hist[4096] = sample size
// refine != nullptr: second level -- only the sample elements of coarse bin refine[2] are counted, in 1024 linear sub-bins of
// that bin's float range (hist[0..1023]); the coarse bins are 0.4 % wide, which on a dense value distribution is far more than
// the quantile margin and would blow the candidate set up
constexpr int SPARTS = 8;
// both selection levels for a 256-thread workgroup (the bodies of eps_select_kernel / eps_select2_kernel below, which stay the two-launch
// API): level 1 = first bin whose cumulative count reaches ceil(q * sample size), threshold = upper edge of the next bin, sel = {threshold
// bits, sample size, bin, sample elements below the bin, target}; level 2 = the sub-bin of coarse bin sel[2] where sel[3] + sub-bins reach
// sel[4], threshold = upper edge of the sub-bin after it.  `part` = 4097 words of LDS scratch.  Histogram words are read with agent-scope loads.
__device__ __forceinline__ void eps_select_block(const unsigned long long* __restrict__ hist, double q, unsigned long long* __restrict__ sel, bool level2,
                                                 unsigned int* part, unsigned long long* __restrict__ split_out = nullptr) {
  const int t = (int)threadIdx.x;
  // the histogram words were produced by device-scope atomics of workgroups on every XCD: read them past this CU's L1 and this XCD's L2
  // (sc0 sc1).  As plain buffer loads, not as atomic loads: hipcc waits for every relaxed atomic load before it issues the next one -- 16
  // dependent memory round trips per thread made the launch that carries the selection 45 us instead of 11 (rocprofv3, N = 16 000)
  typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(hist), 0, 4097 * 8, 0x00020000);
  auto ld = [&](int b) -> unsigned long long {
    const v2u_ x = __builtin_amdgcn_raw_buffer_load_b64(hr, b * 8, 0, 17);
    return (unsigned long long)x[0] | ((unsigned long long)x[1] << 32);
  };
  const int nb = level2 ? 4 : 16, b0 = t * nb;                     // bins per thread: 1024 / 256 or 4096 / 256
  unsigned long long c[16], mine = 0;
#pragma unroll
  for (int u = 0; u < 16; u++) { c[u] = u < nb ? ld(b0 + u) : 0ull; mine += c[u]; }
  // exclusive prefix of the 256 per-thread sums (counts fit 32 bits: a sample holds < 2^32 elements)
  __syncthreads();
  part[t] = (unsigned int)mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned int add = t >= o ? part[t - o] : 0u;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  unsigned long long run = (unsigned long long)part[t] - mine;
  if (!level2) {
    const unsigned long long total = ld(4096);
    unsigned long long target = (unsigned long long)ceil(q * (double)total);
    if (target < 64) target = 64;
    __shared__ int bsel; __shared__ unsigned long long bbefore;
    if (t == 0) { bsel = 4096; bbefore = 0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const unsigned long long before = run;
      run += c[u];
      if (before < target && run >= target) { bsel = b0 + u; bbefore = before; }      // unique
    }
    __syncthreads();
    if (t == 0) {
      const int b = bsel;
      float thr = b >= 4094 ? __uint_as_float(0x7f800000u) : sur_bin_upper(b + 1);
      if (total == 0) thr = __uint_as_float(0x7f800000u);
      sel[0] = (unsigned long long)__float_as_uint(thr); sel[1] = total; sel[2] = (unsigned long long)b;
      sel[3] = b >= 4096 ? 0ull : bbefore; sel[4] = target;
    }
    if (split_out) {
      // splitters of the sample sort that will order the collected keys (round 6: ss_splitters_kernel -- one workgroup sorting a sample
      // of 4096 candidates, 37 us, the longest kernel of the chain -- is not launched): the candidates are the values below the threshold,
      // i.e. the sample mass (0, target]; splitter s sits at mass s * target / 1024, linearly interpolated inside its histogram bin (~6
      // sample elements per bucket: the balance of a sorted sample of 4 keys per bucket).  Any monotone splitters sort correctly -- a bad
      // estimate costs balance (a bucket beyond 16 384 keys raises the sort's fail word and the two-call path answers), never exactness.
      for (int q_ = t; q_ < 1023; q_ += 256) split_out[q_] = ~0ull;       // (a splitter no bin reaches -- a sample below the 64-element floor -- stays +inf)
      __syncthreads();
      unsigned long long before = (unsigned long long)part[t] - mine;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const unsigned long long after = before + c[u];
        if (c[u] && before < target) {
          unsigned long long s_lo = before * 1024ull / target + 1, s_hi = after * 1024ull / target;
          if (s_hi > 1023) s_hi = 1023;
          const int b = b0 + u;
          const double lo = b == 0 ? 0.0 : (double)sur_bin_upper(b - 1), hi = (double)sur_bin_upper(b);
          for (unsigned long long sidx = s_lo; sidx <= s_hi; sidx++) {
            const double m = (double)sidx * (double)target * (1.0 / 1024.0);
            double frac = (m - (double)before) / (double)c[u];
            frac = frac < 0.0 ? 0.0 : (frac > 1.0 ? 1.0 : frac);
            split_out[sidx - 1] = (unsigned long long)__double_as_longlong(lo + frac * (hi - lo));
          }
        }
        before = after;
      }
    }
  } else {
    const int b = (int)sel[2];
    if (b < 1 || b >= 4094) return;                    // degenerate / open-ended bins keep the coarse threshold
    const unsigned long long before0 = sel[3], target = sel[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned long long excl = before0 + run, incl = excl + c[u];
      run += c[u];
      if (excl < target && incl >= target) {
        const float lo = sur_bin_upper(b - 1), hi = sur_bin_upper(b);
        const float thr = lo + (float)(b0 + u + 2) * ((hi - lo) * (1.f / 1024.f));
        sel[0] = (unsigned long long)__float_as_uint(thr < sur_bin_upper(b + 1) ? thr : sur_bin_upper(b + 1));
      }
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void eps_sample_hist_kernel(MatView mv, int stride, const unsigned long long* __restrict__ refine,
                                                              unsigned long long* __restrict__ hist, double q = 0.0, unsigned long long* __restrict__ sel = nullptr,
                                                              unsigned int* __restrict__ ticket = nullptr, unsigned long long* __restrict__ split_out = nullptr) {
  __shared__ unsigned int lh[4097];
  for (int b = (int)threadIdx.x; b < 4097; b += 256) lh[b] = 0;
  __syncthreads();
  const int lane = lane_id();
  const float lam32 = (float)mv.lambda_value;
  const int rbin = refine ? (int)refine[2] : -1;
  const float rlo = rbin > 0 ? sur_bin_upper(rbin - 1) : 0.f, rhi = rbin >= 0 ? sur_bin_upper(rbin) : 1.f;
  const float rinv = 1024.f / (rhi - rlo);
  const int nsamp = (mv.nrows + stride - 1) / stride;
  // a sample row is shared by SPARTS waves (chunks round-robin): 192 rows alone leave the launch at one wave per CU and latency-bound
  for (int widx = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); widx < nsamp * SPARTS; widx += (int)gridDim.x * 4) {
    const int sidx = widx / SPARTS, part = widx - sidx * SPARTS;
    const int il = sidx * stride, gi = mv.row0 + il;
    RowStream rs(il, mv.N, mv.nrows);
    const hbits vi = MODE == 0 ? mv.v[gi] : (hbits)0;
    unsigned cnt = 0;
    for (int c = (rs.first + gi + 1) / 512 + part; c < rs.nchunks; c += SPARTS) {
      float sv[8]; unsigned zm = 0;
      const int j0 = rs.col0(c, lane);
      if (MODE == 2) {
        double dv[8];
        load_vals<MODE>(mv, rs, c, lane, gi, dv);
#pragma unroll
        for (int e = 0; e < 8; e++) { sv[e] = (float)dv[e]; if (dv[e] == 0.0) zm |= 1u << e; }
      } else {
        const RawChunk raw = load_raw<MODE == 2 ? 1 : MODE>(mv, rs, c, lane);
        surrogate8<MODE == 2 ? 1 : MODE>(mv, raw, vi, lam32, sv, zm);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int k = j0 + e;
        if (k > gi && k < mv.N && !((zm >> e) & 1u)) {
          const int b = sur_bin(sv[e]);
          if (!refine) { atomicAdd(&lh[b], 1u); cnt++; }
          else if (b == rbin) {
            int sb = (int)((sv[e] - rlo) * rinv);
            sb = sb < 0 ? 0 : (sb > 1023 ? 1023 : sb);
            atomicAdd(&lh[sb], 1u); cnt++;
          }
        }
      }
    }
    for (int sh = 1; sh < 64; sh <<= 1) cnt += (unsigned)__shfl_xor((int)cnt, sh, 64);
    if (lane == 0 && cnt) atomicAdd(&lh[4096], cnt);
  }
  __syncthreads();
  for (int b = (int)threadIdx.x; b < 4097; b += 256) if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
  if (!sel) return;
  // ---- round 6: the threshold selection rides in this launch.  The workgroup that draws the last ticket has every other workgroup's
  // histogram atomics behind it (they are device-scope read-modify-writes at the memory side; the agent-scope loads of the selection
  // by-pass this CU's L1 and its XCD's L2) and runs what eps_select_kernel / eps_select2_kernel ran as launches of their own.
  // (no cache maintenance is involved: the histogram words are only ever touched by device-scope atomics and agent-scope loads, both
  // performed at the memory side -- a wave waits for the acknowledgement of its own atomics, the barrier collects the waves, lane 0 draws
  // the ticket.  __threadfence() here -- an L2 write-back + invalidate per thread -- cost more than the launches it saved.)
  __shared__ unsigned int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  eps_select_block(hist, q, sel, refine != nullptr, lh, refine ? nullptr : split_out);
}

// one workgroup: first bin whose cumulative sample count reaches ceil(q * sample size); threshold = upper edge of the NEXT
// bin (one guard bin = 0.4 % in value).  out3 = {threshold as float bits, sample size, bin}; degenerate sample -> +inf threshold
__global__ __launch_bounds__(1024) void eps_select_kernel(const unsigned long long* __restrict__ hist, double q, unsigned long long* __restrict__ out3) {
  __shared__ unsigned long long part[1024];
  const int t = (int)threadIdx.x;
  unsigned long long c[4], s = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) { c[u] = hist[4 * t + u]; s += c[u]; }
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {             // Hillis-Steele inclusive scan
    const unsigned long long add = t >= o ? part[t - o] : 0ull;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const unsigned long long total = hist[4096];
  unsigned long long target = (unsigned long long)ceil(q * (double)total);
  if (target < 64) target = 64;
  unsigned long long run = part[t] - s;              // exclusive prefix of this thread's four bins
  __shared__ int bsel;
  if (t == 0) bsel = 4096;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const unsigned long long before = run;
    run += c[u];
    if (before < target && run >= target) bsel = 4 * t + u;   // unique
  }
  __syncthreads();
  if (t == 0) {
    const int b = bsel;
    float thr = b >= 4094 ? __uint_as_float(0x7f800000u) : sur_bin_upper(b + 1);
    if (total == 0) thr = __uint_as_float(0x7f800000u);
    out3[0] = (unsigned long long)__float_as_uint(thr); out3[1] = total; out3[2] = (unsigned long long)b;
  }
  // for the refinement level: sample elements below the selected bin, and the target rank
#pragma unroll
  for (int u = 0; u < 4; u++) if (4 * t + u == bsel) out3[3] = part[t] - s + (u > 0 ? c[0] : 0) + (u > 1 ? c[1] : 0) + (u > 2 ? c[2] : 0);
  if (t == 0) { out3[4] = target; if (bsel >= 4096) out3[3] = 0; }
}

// level 2: hist2 = 1024 sub-bins of coarse bin sel[2]; threshold = upper edge of the sub-bin AFTER the one where the cumulative
// count (sel[3] + sub-bins) reaches the target sel[4] (one guard sub-bin, 4e-6 relative: ten times the surrogate error);
// overwrites sel[0] when the bin was usable
__global__ __launch_bounds__(1024) void eps_select2_kernel(const unsigned long long* __restrict__ hist2, unsigned long long* __restrict__ sel) {
  __shared__ unsigned long long part[1024];
  const int t = (int)threadIdx.x;
  const unsigned long long c = hist2[t];
  part[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long add = t >= o ? part[t - o] : 0ull;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const int b = (int)sel[2];
  if (b < 1 || b >= 4094) return;                    // degenerate / open-ended bins keep the coarse threshold
  const unsigned long long before = sel[3], target = sel[4];
  const unsigned long long incl = before + part[t], excl = incl - c;
  if (excl < target && incl >= target) {
    const float lo = sur_bin_upper(b - 1), hi = sur_bin_upper(b);
    const float thr = lo + (float)(t + 2) * ((hi - lo) * (1.f / 1024.f));
    sel[0] = (unsigned long long)__float_as_uint(thr < sur_bin_upper(b + 1) ? thr : sur_bin_upper(b + 1));
  }
}

// the one full pass: exact float64 keys of the strict-upper non-zero elements whose surrogate is < *thr -> buf (cursor[0]);
// cursor[1] += number of exact zeros in the strict upper triangle
template <int MODE>
__global__ __launch_bounds__(256) void eps_compact_thr_kernel(MatView mv, const unsigned long long* __restrict__ thr3, unsigned long long* __restrict__ buf,
                                                              unsigned long long cap, unsigned long long* __restrict__ cursor,
                                                              const unsigned long long* __restrict__ gate, const unsigned char* __restrict__ rowmask) {
  // key stage (768 keys per wave: the generic path appends up to 512 at once) + candidate-column stage of the fast path (see below):
  // 33.4 KB per workgroup, four workgroups per CU
  constexpr int KCAP = 768, CCAP = 64 + 512;
#ifndef SSG_DENSE_PIPE
#define SSG_DENSE_PIPE 8
#endif
  constexpr int DPIPE = SSG_DENSE_PIPE;          // chunks in flight per wave in the fast path
  __shared__ unsigned long long sbuf[4][KCAP];
  __shared__ unsigned int cidx_s[4][CCAP];
  if (gate && *gate == 0ull) return;          // the sparse pass queued in front of this launch has done every row
  const int lane = lane_id();
  WaveStage<unsigned long long> st{sbuf[threadIdx.x >> 6], 0};
  unsigned int* cidx = cidx_s[threadIdx.x >> 6];
  int nci = 0;                                 // staged candidate columns of the current row (wave-uniform)
  const float thr = __uint_as_float((unsigned)thr3[0]);
  const float lam32 = (float)mv.lambda_value;
  unsigned long long zeros = 0;
  // generic chunk: per-element bounds, exact zero detection (row edges, unaligned rows, modes 1 / 2)
  auto generic_chunk = [&](const RowStream& rs, int c, int gi, hbits vi, const RawChunk* rawp) {
    const int j0 = rs.col0(c, lane);
    float sv[8]; unsigned zm = 0;
    double dv[8];
    RawChunk raw;
    if (MODE == 2) {
      load_vals<MODE>(mv, rs, c, lane, gi, dv);
#pragma unroll
      for (int e = 0; e < 8; e++) { sv[e] = (float)dv[e]; if (dv[e] == 0.0) zm |= 1u << e; }
    } else {
      raw = rawp ? *rawp : load_raw<MODE == 2 ? 1 : MODE>(mv, rs, c, lane);
      surrogate8<MODE == 2 ? 1 : MODE>(mv, raw, vi, lam32, sv, zm);
    }
    unsigned cand = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int k = j0 + e;
      if (k > gi && k < mv.N) {
        if ((zm >> e) & 1u) zeros++;
        else if (sv[e] < thr) cand |= 1u << e;
      }
    }
    if (!__any(cand != 0)) return;
    if (MODE != 2) decode_raw<MODE == 2 ? 1 : MODE>(mv, raw, vi, dv);   // exact float64 values of this chunk (rare path)
    unsigned long long keys[8]; int n = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      keys[e] = ~0ULL;
      if ((cand >> e) & 1u) {
        if (dv[e] != 0.0) { keys[e] = (unsigned long long)__double_as_longlong(dv[e]); n++; }
        else zeros++;
      }
    }
    int incl = n;
    for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_up(incl, sh, 64); if (lane >= sh) incl += o; }
    const int tot = __shfl(incl, 63, 64);
    int w = st.n + incl - n;
#pragma unroll
    for (int e = 0; e < 8; e++) if (keys[e] != ~0ULL) st.buf[w++] = keys[e];
    st.n += tot;
    if (st.n > KCAP - 512) st.flush(buf, cap, cursor, lane);
  };
  // the top `m` (<= 64) staged candidates (J' half | v_k half << 16) of a row -> exact float64 keys (exact zeros are counted, not kept) -> key stage
  auto finish = [&](const hbits* __restrict__ Mrow, hbits vi, int m) {
    (void)Mrow;
    unsigned long long key = ~0ULL; bool keep = false;
    if (lane < m) {
      const unsigned pk = cidx[nci - m + lane];
      const double d = final_dist_value((hbits)(pk & 0xffffu), vi, (hbits)(pk >> 16), mv.lambda_value);
      if (d != 0.0) { key = (unsigned long long)__double_as_longlong(d); keep = true; }
      else zeros++;
    }
    const uint64_t bm = __ballot(keep);
    if (keep) st.buf[st.n + __popcll(bm & lanemask_lt())] = key;
    st.n += __popcll(bm);
    nci -= m;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (st.n > KCAP - 512) st.flush(buf, cap, cursor, lane);
  };
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const bool fast_ok = MODE == 0 && (mv.N & 7) == 0 && thr > 0.f;
  // rows of the strict upper triangle get shorter linearly: a wave takes rows in complementary pairs (p, nrows-1-p), so every
  // wave streams the same number of elements (a handful of whole rows per wave would leave a 1.5-2x imbalance)
  const int npairs = (mv.nrows + 1) / 2;
  for (int pidx = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); pidx < npairs; pidx += (int)gridDim.x * 4)
  for (int hh = 0; hh < 2; hh++) {
    const int il = hh ? mv.nrows - 1 - pidx : pidx;
    if (hh && il == pidx) continue;                // odd row count: the middle row is its own partner
    if (rowmask && rowmask[il] == 0) continue;     // done through the sparse copy
    const int gi = mv.row0 + il;
    RowStream rs(il, mv.N, mv.nrows);
    const int c0 = (rs.first + gi + 1) / 512;
    const hbits vi = MODE == 0 ? mv.v[gi] : (hbits)0;
    if (fast_ok) {
      // N % 8 == 0: rows are 16-byte aligned.  The chunk holding the diagonal and a partial last chunk take the generic path;
      // every chunk in between lies entirely in the strict upper triangle: no bounds, packed half adds, and no zero test --
      // an exact zero has surrogate 0 (or J' alone when lambda == 0) < thr, so it is a candidate and is classified exactly below
      const hbits* M = reinterpret_cast<const hbits*>(mv.M) + (int64_t)il * mv.N;
      const _Float16 vi16 = __builtin_bit_cast(_Float16, vi);
      const h2 vi2 = {vi16, vi16};
      const int nfull = mv.N / 512;
      if (c0 < rs.nchunks) generic_chunk(rs, c0, gi, vi, nullptr);
      for (int cb = c0 + 1; cb < nfull; cb += DPIPE) {
        uint4 xjs[DPIPE], xvs[DPIPE];
#pragma unroll
        for (int u = 0; u < DPIPE; u++) {
          const int j0 = min(cb + u, nfull - 1) * 512 + lane * 8;
          xjs[u] = *reinterpret_cast<const uint4*>(M + j0);
          xvs[u] = *reinterpret_cast<const uint4*>(mv.v + j0);
        }
#pragma unroll
        for (int u = 0; u < DPIPE; u++) {
          const int c = cb + u;
          if (c >= nfull) break;
          const unsigned wj[4] = {xjs[u].x, xjs[u].y, xjs[u].z, xjs[u].w}, wv[4] = {xvs[u].x, xvs[u].y, xvs[u].z, xvs[u].w};
          unsigned cand = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const h2 s2 = __builtin_bit_cast(h2, wv[q]) + vi2;                 // half(v_k + v_i), two columns per instruction
            const h2 j2 = __builtin_bit_cast(h2, wj[q]);
            const float a = (float)j2.x + (float)s2.x * lam32, b = (float)j2.y + (float)s2.y * lam32;
            cand |= (a < thr ? 1u : 0u) << (2 * q) | (b < thr ? 1u : 0u) << (2 * q + 1);
          }
          // Round 6: at the 1.3 rho quantile about two chunks in three hold a candidate, so the "rare" path ran for most chunks -- and it
          // rebuilt the float64 value of all 8 x 64 elements of the chunk for that one candidate (the pass streamed at 2.4-2.9 TB/s
          // against 4.5-5.1 with a threshold nothing lies below).  Now a chunk only stages the two HALVES of each candidate (J' and v_k);
          // the exact float64 keys are built 64 candidates at a time, one per lane (`finish`).
          const int n = __popc(cand);
          const uint64_t bm = __ballot(n != 0);
          if (bm == 0) continue;
          // (staged: the two halves the key is made of -- J' and v_k --, not the column: `finish` needs no load)
          if (!__any(n > 1)) {
            // the usual case, one candidate in each of a few lanes: the slot is the lane's rank among them (two mbcnt instructions, no scan)
            if (n) {
              const int e = __ffs((int)cand) - 1, sh = (e & 1) * 16;
              const unsigned pj = e < 4 ? (e < 2 ? wj[0] : wj[1]) : (e < 6 ? wj[2] : wj[3]), pv = e < 4 ? (e < 2 ? wv[0] : wv[1]) : (e < 6 ? wv[2] : wv[3]);
              cidx[nci + __popcll(bm & lanemask_lt())] = ((pj >> sh) & 0xffffu) | (((pv >> sh) & 0xffffu) << 16);
            }
            nci += __popcll(bm);
          } else {
            int incl = n;
            for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_up(incl, sh, 64); if (lane >= sh) incl += o; }
            const int tot = __shfl(incl, 63, 64);
            int w = nci + incl - n;
#pragma unroll
            for (int e = 0; e < 8; e++)
              if ((cand >> e) & 1u) cidx[w++] = ((wj[e >> 1] >> ((e & 1) * 16)) & 0xffffu) | (((wv[e >> 1] >> ((e & 1) * 16)) & 0xffffu) << 16);
            nci += tot;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          while (nci >= 64) finish(M, vi, 64);
        }
      }
      if (nci) finish(M, vi, nci);                      // the row's last few candidates (vi is the row's)
      if (nfull * 512 < mv.N && nfull > c0) generic_chunk(rs, nfull, gi, vi, nullptr);
      continue;
    }
    for (int c = c0; c < rs.nchunks; c++) generic_chunk(rs, c, gi, vi, nullptr);
  }
  block_flush(st, buf, cap, cursor);
  for (int sh = 1; sh < 64; sh <<= 1) zeros += (unsigned long long)__shfl_xor((long long)zeros, sh, 64);
  if (lane == 0 && zeros) atomicAdd(&cursor[1], zeros);
}

__global__ void fill_u64_kernel(unsigned long long* p, unsigned long long n0, unsigned long long n1, unsigned long long v) {
  for (unsigned long long i = n0 + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n1; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------ bitonic sort (u64, ascending)
constexpr int SORT_CH = 2048;
__device__ __forceinline__ void cswap(unsigned long long& a, unsigned long long& b, bool up) {
  if ((a > b) == up) { const unsigned long long t = a; a = b; b = t; }
}
// k_lo..k_hi: run all (k, j) stages with j < SORT_CH inside LDS.  For k <= SORT_CH this is a full
// local sort; for k > SORT_CH only the tail j = SORT_CH/2 .. 1 of stage k.
__global__ __launch_bounds__(1024) void bitonic_local_kernel(unsigned long long* __restrict__ a, unsigned long long k_first, unsigned long long k_last) {
  __shared__ unsigned long long s[SORT_CH];
  const unsigned long long g0 = (unsigned long long)blockIdx.x * SORT_CH;
  const int t = (int)threadIdx.x;
  s[t] = a[g0 + t]; s[t + 1024] = a[g0 + t + 1024];
  __syncthreads();
  for (unsigned long long k = k_first; k <= k_last; k <<= 1) {
    for (unsigned long long j = (k >> 1) < (unsigned long long)(SORT_CH / 2) ? (k >> 1) : (unsigned long long)(SORT_CH / 2); j > 0; j >>= 1) {
      // thread t handles the pair (i, i^j) with i = the t-th index whose j bit is clear
      const unsigned long long i = ((unsigned long long)t / j) * (2 * j) + ((unsigned long long)t % j);
      const bool up = (((g0 + i) & k) == 0);
      unsigned long long x = s[i], y = s[i + j];
      cswap(x, y, up);
      s[i] = x; s[i + j] = y;
      __syncthreads();
    }
  }
  a[g0 + t] = s[t]; a[g0 + t + 1024] = s[t + 1024];
}
// NS consecutive global stages (strides j, j/2, .. j >> (NS-1), all >= SORT_CH) of merge level k in ONE launch: a thread holds the 2^NS
// elements whose indices differ only in those stride bits, so every compare-exchange of the NS stages stays inside its registers
// (round 4: one launch per stage was 36 launches of ~3 us for 2^19 keys; three stages per launch make it 15).  k >= 2j, so the
// sort direction is the same for all of a thread's elements.
template <int NS>
__global__ __launch_bounds__(256) void bitonic_global_kernel(unsigned long long* __restrict__ a, unsigned long long n, unsigned long long k, unsigned long long j) {
  constexpr int NE = 1 << NS;
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / NE) return;
  const unsigned long long jl = j >> (NS - 1);                  // smallest stride of this launch
  const unsigned long long base = (t / jl) * (NE * jl) + (t % jl);
  const bool up = ((base & k) == 0);
  unsigned long long e[NE];
#pragma unroll
  for (int b = 0; b < NE; b++) e[b] = a[base + (unsigned long long)b * jl];
#pragma unroll
  for (int s = NS - 1; s >= 0; s--) {                           // stride jl << s
#pragma unroll
    for (int b = 0; b < NE; b++)
      if (!(b & (1 << s))) cswap(e[b], e[b | (1 << s)], up);
  }
#pragma unroll
  for (int b = 0; b < NE; b++) a[base + (unsigned long long)b * jl] = e[b];
}

// ------------------------------------------------------------------ numpy pairwise mean
// np.mean(x[:top]) = pairwise_sum(x, top) / top with numpy's recursion (blocks <= 128 summed
// with 8 accumulators, halves split at a multiple of 8).  The recursion tree depends on `top`
// only, so the host enumerates it once (leaves + internal nodes grouped by height) and one
// workgroup evaluates it: all leaves in parallel, then one tree level per barrier.
template <typename T>
__device__ T pw_leaf(const unsigned long long* keys, long long off, int n) {
  auto val = [&](long long i) -> T { return (T)__longlong_as_double((long long)keys[off + i]); };
  if (n < 8) { T r = (T)0; for (int i = 0; i < n; i++) r += val(i); return r; }
  T r[8]; int i;
  for (i = 0; i < 8; i++) r[i] = val(i);
  for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += val(i + j);
  T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += val(i);
  return res;
}
// the a-posteriori checks of the sampled eps rule (see eps_check_kernel, which runs them as a launch of its own): one thread
struct EpsCheckArgs {
  const unsigned long long* cursor = nullptr; const unsigned long long* thr3 = nullptr; double rho = 0.0; unsigned long long upper_total = 0; long long top_guess = 0;
  unsigned long long n_cap = 0; unsigned long long* status6 = nullptr; const unsigned long long* sort_fail = nullptr;
};
__device__ __forceinline__ void eps_check_one(const unsigned long long* __restrict__ sorted, const EpsCheckArgs& a, double* __restrict__ eps2) {
  const unsigned long long got = a.cursor[0], zeros = a.cursor[1];
  const long long count = (long long)(a.upper_total - zeros);
  const long long top = (long long)rint(a.rho * (double)count);
  const double thr = (double)__uint_as_float((unsigned)(a.thr3[0] & 0xffffffffull));
  bool ok = top == a.top_guess && top > 0 && got <= a.n_cap && got >= (unsigned long long)top && isfinite(thr) && !(a.sort_fail && *a.sort_fail);
  unsigned long long kb = 0;
  if (ok) {
    kb = sorted[top - 1];
    const double key_top = __longlong_as_double((long long)kb);
    ok = key_top < thr - 1e-6 * (1.0 + fabs(thr));
  }
  a.status6[0] = ok ? 1ull : 0ull; a.status6[1] = got; a.status6[2] = zeros; a.status6[3] = (unsigned long long)top; a.status6[4] = kb; a.status6[5] = a.thr3[0];
  if (!ok) eps2[0] = __longlong_as_double(0x7ff8000000000000ll);
}
// leaves of the pairwise tree, 8 lanes per leaf (one lane per accumulator of numpy's unrolled loop), 8 leaves per wave:
//   r[j] = a[j]; for i = 8, 16, ...: r[j] += a[i + j];  res = ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7));  res += a[tail...]
template <typename T>
__global__ __launch_bounds__(256) void eps_leaf_kernel(const unsigned long long* __restrict__ keys, int nleaves, const long long* __restrict__ leaf_off,
                                                       const int* __restrict__ leaf_n, T* __restrict__ val) {
  const int gl = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);      // leaf
  const int j = (int)(threadIdx.x & 7);
  const bool live = gl < nleaves;
  const long long off = live ? leaf_off[gl] : 0;
  const int n = live ? leaf_n[gl] : 0;
  auto a = [&](int i) -> T { return (T)__longlong_as_double((long long)keys[off + i]); };
  T res;
  if (n < 8) {
    res = (T)0;
    if (j == 0) for (int i = 0; i < n; i++) res += a(i);
  } else {
    const int nb = n - (n % 8);
    T r = a(j);
    for (int i = 8; i < nb; i += 8) r += a(i + j);
    // combine in numpy's order: lanes 2q, 2q+1 -> pairs, then pairs of pairs, then the two halves (xor shuffles inside the 8 lanes)
    T p = r + (T)__shfl_xor(r, 1, 64);         // lane even: r_j + r_{j+1} (addition is commutative, the operand order does not matter)
    T q = p + (T)__shfl_xor(p, 2, 64);         // lanes 0-3: (r0+r1)+(r2+r3); lanes 4-7: (r4+r5)+(r6+r7)
    res = q + (T)__shfl_xor(q, 4, 64);
    if (j == 0) for (int i = nb; i < n; i++) res += a(i);
  }
  if (live && j == 0) val[gl] = res;
}
// out[0] = eps as double; out[1] = half bits of eps (mode 1) as a double-held integer
template <typename T>
__global__ __launch_bounds__(1024) void eps_mean_kernel(const unsigned long long* __restrict__ keys, long long top, int nleaves, int nlevels,
                                                        const long long* __restrict__ leaf_off, const int* __restrict__ leaf_n,
                                                        const int* __restrict__ node_l, const int* __restrict__ node_r,
                                                        const int* __restrict__ level_ptr, T* __restrict__ val, double* __restrict__ out, EpsCheckArgs chk = EpsCheckArgs()) {
  (void)leaf_off; (void)leaf_n;        // the leaves were summed by eps_leaf_kernel
  for (int h = 0; h < nlevels; h++) {
    for (int x = level_ptr[h] + (int)threadIdx.x; x < level_ptr[h + 1]; x += (int)blockDim.x) val[nleaves + x] = val[node_l[x]] + val[node_r[x]];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int ninternal = level_ptr[nlevels];
    const T s = val[ninternal ? nleaves + ninternal - 1 : 0];   // the root is the last node of the last level
    if (sizeof(T) == 8) { out[0] = (double)s / (double)top; out[1] = 0.0; }
    else {
      const hbits e = d2h((double)s / (double)top);   // np.float32 scalar / np.intp -> float64 -> np.float16
      out[0] = (double)h2f(e); out[1] = (double)e;
    }
    if (chk.status6) eps_check_one(keys, chk, out);     // round 6: the a-posteriori checks in the same launch (ssg_eps_mean_check)
  }
}

// ------------------------------------------------------------------ K11 region query
// cnt[il] = #{k : d(i,k) <= eps} (self included when d(i,i) <= eps);  edges (i,k) incl. self hits.
struct Edge { int i, k; };

// append the hit columns (bit e of hitmask -> column j0+e) of this wave to its LDS stage (out of line: rare)
__device__ __forceinline__ int rq_append(WaveStage<Edge>& st, unsigned hitmask, int j0, int gi, Edge* eout, unsigned long long cap,
                                                   unsigned long long* cursor) {
  const int lane = lane_id();
  const uint64_t lt = lanemask_lt();
  int added = 0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const bool h = (hitmask >> e) & 1u;
    const uint64_t m = __ballot(h);
    if (h) { Edge& x = st.buf[st.n + added + __popcll(m & lt)]; x.i = gi; x.k = j0 + e; }
    added += __popcll(m);
  }
  st.n += added;
  if (st.n > STAGE_CAP - 512) st.flush(eout, cap, cursor, lane);
  return added;
}

// generic chunk (row edges, unaligned rows, plain matrices): exact value of every element
template <int MODE>
__device__ __forceinline__ unsigned rq_generic_chunk(const MatView& mv, const RowStream& rs, int c, int lane, int gi, double eps) {
  double dv[8];
  load_vals<MODE>(mv, rs, c, lane, gi, dv);
  const int j0 = rs.col0(c, lane);
  unsigned hitmask = 0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = j0 + e;
    if (k >= 0 && k < mv.N && dv[e] <= eps) hitmask |= 1u << e;
  }
  return hitmask;
}

// exact decision for the elements flagged by the packed-half prefilter (mode 0)
__device__ __forceinline__ unsigned rq_decide(const MatView& mv, unsigned flags, const uint4& xj, const uint4& xv, int gi, double eps) {
  const unsigned wj[4] = {xj.x, xj.y, xj.z, xj.w}, wv[4] = {xv.x, xv.y, xv.z, xv.w};
  const hbits vi = mv.v[gi];
  unsigned hitmask = 0;
  while (flags) {
    const int e = __ffs((int)flags) - 1;
    flags &= flags - 1;
    const unsigned a = e < 2 ? wj[0] : e < 4 ? wj[1] : e < 6 ? wj[2] : wj[3], b = e < 2 ? wv[0] : e < 4 ? wv[1] : e < 6 ? wv[2] : wv[3];
    const hbits jp = (hbits)((a >> ((e & 1) * 16)) & 0xffffu), vk = (hbits)((b >> ((e & 1) * 16)) & 0xffffu);
    if (final_dist_value(jp, vi, vk, mv.lambda_value) <= eps) hitmask |= 1u << e;    // exact, rerank.py:122
  }
  return hitmask;
}

template <int MODE>
__global__ __launch_bounds__(256) void region_query_kernel(MatView mv, double eps_arg, int32_t* __restrict__ cnt, int32_t* __restrict__ edges,
                                                           unsigned long long cap, unsigned long long* __restrict__ cursor,
                                                           const unsigned long long* __restrict__ gate, const unsigned char* __restrict__ rowmask,
                                                           const double* __restrict__ eps_dev) {
  __shared__ Edge sbuf[4][STAGE_CAP];
  if (gate && *gate == 0ull) return;          // the sparse pass queued in front of this launch has done every row
  const double eps = eps_dev ? *eps_dev : eps_arg;     // round 5: eps left on the device by the eps rule queued in front (NaN = "no eps": no hit)
  const int lane = lane_id();
  WaveStage<Edge> st{sbuf[threadIdx.x >> 6], 0};
  Edge* eout = reinterpret_cast<Edge*>(edges);
  // mode-0 prefilter in packed half math (2 elements per instruction): d16 = half(v_i+v_k)*half(lambda) + J' is
  // within `band` of the exact float64 value, so only elements with d16 < eps + band can be hits; those (a
  // handful per row) are decided exactly.  half(v_i + v_k) with the native half add equals numpy's
  // float32-add-then-round for every pair of halves.
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const float bandf = 0.0078125f * (1.f + fabsf((float)eps) + fabsf((float)mv.lambda_value));   // >= 2^-9*lambda*|s| + 3*2^-11*|d| with margin
  const _Float16 thr16 = (_Float16)((float)eps + bandf + 0.002f * (1.f + fabsf((float)eps)));   // margin also covers the rounding of the threshold itself
  const h2 thr2 = {thr16, thr16};
  const h2 lam2 = {(_Float16)(float)mv.lambda_value, (_Float16)(float)mv.lambda_value};
  const bool fast_ok = MODE == 0 && (mv.N & 7) == 0 && (float)eps < 30000.f && fabsf((float)mv.lambda_value) < 16.f;
  for (int il = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); il < mv.nrows; il += (int)gridDim.x * 4) {
    if (rowmask && rowmask[il] == 0) continue;     // done through the sparse copy
    const int gi = mv.row0 + il;
    RowStream rs(il, mv.N, mv.nrows);
    int rowcnt = 0;
    if (fast_ok) {
      // N % 8 == 0: every row starts on a 16-byte boundary, all chunks are aligned; the last one may be partial
      const hbits* M = reinterpret_cast<const hbits*>(mv.M) + (int64_t)il * mv.N;
      const _Float16 vi16 = __builtin_bit_cast(_Float16, mv.v[gi]);
      const h2 vi2 = {vi16, vi16};
      const int nfull = mv.N / 512;
      for (int cb = 0; cb < nfull; cb += PIPE) {
        uint4 xjs[PIPE], xvs[PIPE];
#pragma unroll
        for (int u = 0; u < PIPE; u++) {
          const int j0 = min(cb + u, nfull - 1) * 512 + lane * 8;       // clamped: the loads of a short tail are repeated, never out of range
          xjs[u] = *reinterpret_cast<const uint4*>(M + j0);
          xvs[u] = *reinterpret_cast<const uint4*>(mv.v + j0);
        }
#pragma unroll
        for (int u = 0; u < PIPE; u++) {
          const int c = cb + u;
          if (c >= nfull) break;
          const int j0 = c * 512 + lane * 8;
          const uint4 xj = xjs[u], xv = xvs[u];
          const unsigned wj[4] = {xj.x, xj.y, xj.z, xj.w}, wv[4] = {xv.x, xv.y, xv.z, xv.w};
          unsigned flags = 0;   // bit e set: element e may be <= eps
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const h2 s2 = __builtin_bit_cast(h2, wv[q]) + vi2;
            const h2 d2 = s2 * lam2 + __builtin_bit_cast(h2, wj[q]);
            const unsigned sg = __builtin_bit_cast(unsigned, (h2)(d2 - thr2));     // sign bit set <=> d16 < thr (NaNs never hit)
            flags |= ((sg >> 15) & 1u) << (2 * q) | ((sg >> 31) & 1u) << (2 * q + 1);
          }
          if (!__any(flags != 0)) continue;                         // the common case: one branch per KiB
          const unsigned hitmask = rq_decide(mv, flags, xj, xv, gi, eps);
          if (__any(hitmask != 0)) rowcnt += rq_append(st, hitmask, j0, gi, eout, cap, cursor);
        }
      }
      if (nfull * 512 < mv.N) {
        const unsigned hitmask = rq_generic_chunk<MODE>(mv, rs, nfull, lane, gi, eps);
        if (__any(hitmask != 0)) rowcnt += rq_append(st, hitmask, rs.col0(nfull, lane), gi, eout, cap, cursor);
      }
    } else {
      for (int c = 0; c < rs.nchunks; c++) {
        const unsigned hitmask = rq_generic_chunk<MODE>(mv, rs, c, lane, gi, eps);
        if (__any(hitmask != 0)) rowcnt += rq_append(st, hitmask, rs.col0(c, lane), gi, eout, cap, cursor);
      }
    }
    if (lane == 0) cnt[il] = rowcnt;
  }
  block_flush(st, eout, cap, cursor);
}

// ------------------------------------------------------------------ round 4: the sparse copy S of J' (jaccard.hip, second generation)
// S holds, per (row, column chunk), the packed words (J' << 17 | column) of every column the row's Jaccard walk touched; every other
// column of the row holds the constant J'(0) = half(1 - lambda), the LARGEST value J' takes.  final_dist = J' + lambda * half(v_i + v_k)
// with v >= 0 and lambda >= 0, so an entry of row i outside S is >= J'(0) + lambda * half(v_i + min_k v_k) =: floor_i (every operation in
// that expression is monotone in v_k): while the bound a pass asks about stays below floor_i, only S can hold what it looks for in
// row i -- a few hundred entries instead of N.  The decision is taken PER ROW on the device: rows whose floor is too low (and all
// rows when S is incomplete) are flagged in a row mask and done by the dense pass queued behind the sparse one.
constexpr int SBATCH = 4;
constexpr int SPARSE_STAGE = 256;        // per-wave staging entries of the sparse passes (a row yields a handful of keys / edges)
struct SparseView {
  const uint32_t* pool; const int64_t* seg_off; const int32_t* seg_len; int nseg;
  const unsigned long long* s_cursor;      // [1] != 0: a segment did not fit, S is unusable
  const uint32_t* vmin;                    // half bits of min_k v_k
  hbits jp0;                               // J'(0)
};
// Can S be used at all?  (all lanes agree)
__device__ __forceinline__ bool sparse_usable(const SparseView& sv, const MatView& mv) {
  return sv.pool && mv.mode == 0 && sv.s_cursor[1] == 0ull && mv.lambda_value >= 0.0 && (sv.jp0 & 0x7fffu) != 0 && (sv.jp0 & 0x7fffu) < 0x7c00u &&
         (sv.jp0 & 0x8000u) == 0;
}

// eps rule, the one full pass, on S: exact float64 keys of the strict-upper non-zero elements whose surrogate is < *thr -> buf
// (cursor[0]); cursor[1] += exact zeros.  A row is walked through S when no column outside S can be a candidate: the surrogate of such
// a column is >= J'(0) + half(v_i + vmin) * lambda in the very float operations the dense pass uses, so the test is exact.  Rows that
// fail it (and every row when S is unusable) get rowmask = 1 and raise cursor[2]: the dense pass queued behind this launch does them.
__global__ __launch_bounds__(256) void eps_compact_sparse_kernel(MatView mv, SparseView sv, const unsigned long long* __restrict__ thr3,
                                                                 unsigned long long* __restrict__ buf, unsigned long long cap,
                                                                 unsigned long long* __restrict__ cursor, unsigned char* __restrict__ rowmask) {
  __shared__ unsigned long long sbuf[4][SPARSE_STAGE];
  const int lane = lane_id();
  const float thr = __uint_as_float((unsigned)thr3[0]);
  const bool usable = thr > 0.f && sparse_usable(sv, mv);
  const hbits vmin = usable ? (hbits)*sv.vmin : (hbits)0;
  WaveStage<unsigned long long> st{sbuf[threadIdx.x >> 6], 0};
  const float lam32 = (float)mv.lambda_value;
  unsigned long long zeros = 0;
  bool anydense = false;
  for (int il = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); il < mv.nrows; il += (int)gridDim.x * 4) {
    const int gi = mv.row0 + il;
    const hbits vi = mv.v[gi];
    const bool rowok = usable && (h2f(sv.jp0) + h2f(h_add(vmin, vi)) * lam32 >= thr);
    if (lane == 0) rowmask[il] = rowok ? 0 : 1;
    if (!rowok) { anydense = true; continue; }
    for (int sg = 0; sg < sv.nseg; sg++) {
      const int64_t off = sv.seg_off[(int64_t)il * sv.nseg + sg];
      const int len = sv.seg_len[(int64_t)il * sv.nseg + sg];
      // SBATCH entries per lane in flight: the packed words first, then the gathers of v[k] they address (a segment is a few hundred
      // entries: one entry per lane and trip would make the row a chain of L2 round trips)
      for (int q0 = 0; q0 < len; q0 += 64 * SBATCH) {
        uint32_t e[SBATCH]; hbits vk[SBATCH];
#pragma unroll
        for (int u = 0; u < SBATCH; u++) { const int q = q0 + u * 64 + lane; e[u] = sv.pool[off + (q < len ? q : len - 1)]; }
#pragma unroll
        for (int u = 0; u < SBATCH; u++) vk[u] = mv.v[e[u] & 0x1ffffu];
#pragma unroll
        for (int u = 0; u < SBATCH; u++) {
          const int q = q0 + u * 64 + lane;
          if (q0 + u * 64 >= len) break;
          unsigned long long key = ~0ULL;
          const int k = (int)(e[u] & 0x1ffffu);
          if (q < len && k > gi) {
            const hbits jp = (hbits)(e[u] >> 17);
            const float sur = h2f(jp) + h2f(h_add(vk[u], vi)) * lam32;
            if (sur < thr) {           // (an exact zero has surrogate 0 < thr: it is always classified here)
              const double d = final_dist_value(jp, vi, vk[u], mv.lambda_value);
              if (d != 0.0) key = (unsigned long long)__double_as_longlong(d); else zeros++;
            }
          }
          const uint64_t bm = __ballot(key != ~0ULL);
          if (bm) {
            if (key != ~0ULL) st.buf[st.n + __popcll(bm & lanemask_lt())] = key;
            st.n += __popcll(bm);
            if (st.n > SPARSE_STAGE - 64) st.flush(buf, cap, cursor, lane);
          }
        }
      }
    }
  }
  block_flush(st, buf, cap, cursor);
  for (int sh = 1; sh < 64; sh <<= 1) zeros += (unsigned long long)__shfl_xor((long long)zeros, sh, 64);
  if (lane == 0 && zeros) atomicAdd(&cursor[1], zeros);
  if (lane == 0 && anydense) cursor[2] = 1ull;
}

// region query on S: row i is walked through S when floor_i = f64(J'(0)) + f64(half(v_i + vmin)) * lambda > eps (no column outside S can
// be a neighbour then); other rows (all rows when S is unusable) get rowmask = 1 and raise cursor[1] for the dense pass behind this launch
__global__ __launch_bounds__(256) void region_query_sparse_kernel(MatView mv, SparseView sv, double eps_arg, int32_t* __restrict__ cnt, int32_t* __restrict__ edges,
                                                                  unsigned long long cap, unsigned long long* __restrict__ cursor,
                                                                  unsigned char* __restrict__ rowmask, const double* __restrict__ eps_dev) {
  __shared__ Edge sbuf[4][SPARSE_STAGE];
  const double eps = eps_dev ? *eps_dev : eps_arg;
  const int lane = lane_id();
  WaveStage<Edge> st{sbuf[threadIdx.x >> 6], 0};
  Edge* eout = reinterpret_cast<Edge*>(edges);
  const bool usable = sparse_usable(sv, mv);
  const hbits vmin = usable ? (hbits)*sv.vmin : (hbits)0;
  bool anydense = false;
  for (int il = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); il < mv.nrows; il += (int)gridDim.x * 4) {
    const int gi = mv.row0 + il;
    const hbits vi = mv.v[gi];
    const bool rowok = usable && final_dist_value(sv.jp0, vi, vmin, mv.lambda_value) > eps;
    if (lane == 0) rowmask[il] = rowok ? 0 : 1;
    if (!rowok) { anydense = true; continue; }
    int rowcnt = 0;
    for (int sg = 0; sg < sv.nseg; sg++) {
      const int64_t off = sv.seg_off[(int64_t)il * sv.nseg + sg];
      const int len = sv.seg_len[(int64_t)il * sv.nseg + sg];
      for (int q0 = 0; q0 < len; q0 += 64 * SBATCH) {
        uint32_t e[SBATCH]; hbits vk[SBATCH];
#pragma unroll
        for (int u = 0; u < SBATCH; u++) { const int q = q0 + u * 64 + lane; e[u] = sv.pool[off + (q < len ? q : len - 1)]; }
#pragma unroll
        for (int u = 0; u < SBATCH; u++) vk[u] = mv.v[e[u] & 0x1ffffu];
#pragma unroll
        for (int u = 0; u < SBATCH; u++) {
          const int q = q0 + u * 64 + lane;
          if (q0 + u * 64 >= len) break;
          const int k = (int)(e[u] & 0x1ffffu);
          const bool hit = q < len && final_dist_value((hbits)(e[u] >> 17), vi, vk[u], mv.lambda_value) <= eps;     // exact, rerank.py:122
          const uint64_t bm = __ballot(hit);
          if (bm) {
            if (hit) { Edge ed; ed.i = gi; ed.k = k; st.buf[st.n + __popcll(bm & lanemask_lt())] = ed; }
            st.n += __popcll(bm); rowcnt += __popcll(bm);
            if (st.n > SPARSE_STAGE - 64) st.flush(eout, cap, cursor, lane);
          }
        }
      }
    }
    if (lane == 0) cnt[il] = rowcnt;
  }
  block_flush(st, eout, cap, cursor);
  if (lane == 0 && anydense) cursor[1] = 1ull;
}

// ------------------------------------------------------------------ K12 union-find
__device__ __forceinline__ int uf_load(int* parent, int x) { return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ int uf_find(int* parent, int x) {
  for (;;) {
    const int p = uf_load(parent, x);
    if (p == x) return x;
    const int gp = uf_load(parent, p);
    if (gp != p) atomicMin(&parent[x], gp);   // path halving; parents only ever decrease
    x = p;
  }
}
__device__ void uf_union(int* parent, int a, int b) {
  for (;;) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    if (atomicCAS(&parent[b], b, a) == b) return;   // hook the larger root under the smaller
  }
}
__global__ void cc_init_kernel(const int32_t* __restrict__ cnt, int N, int min_samples, int* __restrict__ parent, int* __restrict__ lab) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < N) { parent[i] = i; lab[i] = 0x7fffffff; (void)cnt; (void)min_samples; }
}
// The region query emits a row's edges (i, k1), (i, k2), ... next to each other, so a wave holds runs of lanes with the same i.  The lanes
// of a run agree on m = min(k1, k2, ...) over the run's core-core edges (a segmented min: 12 shuffles); `first` marks the run's first lane.
__device__ __forceinline__ void cc_run_min(const int32_t* __restrict__ edges, unsigned long long e, unsigned long long ne, const int32_t* __restrict__ cnt,
                                           int min_samples, int lane, int& i, int& k, bool& valid, int& m, bool& first) {
  i = -1 - lane; k = 0x7fffffff; valid = false;                                   // (idle lanes: a run of their own)
  if (e < ne) {
    i = edges[2 * e]; const int kk = edges[2 * e + 1];
    valid = i != kk && cnt[i] >= min_samples && cnt[kk] >= min_samples;
    if (valid) k = kk;
  }
  m = k;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int oi = __shfl_up(i, d, 64), om = __shfl_up(m, d, 64);
    if (lane >= d && oi == i) m = om < m ? om : m;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int oi = __shfl_down(i, d, 64), om = __shfl_down(m, d, 64);
    if (lane + d < 64 && oi == i) m = om < m ? om : m;
  }
  const int pi = __shfl_up(i, 1, 64);
  first = lane == 0 || pi != i;
}
// Pass 1 (round 5): every core point is hooked under its smallest core neighbour with ONE fire-and-forget atomicMin per run -- no find,
// no compare-and-swap, no dependent chain of device-scope accesses (each is ~1.5 us on this part: the L2s of the 8 XCDs are not coherent
// with each other, a device-scope access goes out to the fabric).  parent[i] <= i stays a valid forest inside i's component; a clique --
// what a DBSCAN cluster mostly is -- ends with every member pointing at its smallest one.
__global__ void cc_hook_kernel(const int32_t* __restrict__ edges, unsigned long long ne, const unsigned long long* __restrict__ ne_dev,
                               const int32_t* __restrict__ cnt, int min_samples, int* __restrict__ parent) {
  if (ne_dev) { const unsigned long long d = *ne_dev; ne = d < ne ? d : ne; }     // device-side count (region query cursor), capped by the capacity
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long eb = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x - lane; eb < ne; eb += stride) {     // (wave-uniform trip count)
    int i, k, m; bool valid, first;
    cc_run_min(edges, eb + lane, ne, cnt, min_samples, lane, i, k, valid, m, first);
    if (first && m < i) atomicMin(&parent[i], m);                                  // (m < i implies the run holds a core-core edge, so i is core)
  }
}
// Pass 2: the union-find proper, for the edges pass 1 left open.  Two rounds of plain loads (an ancestor read a moment too early is still
// an ancestor in the same component) show for nearly every edge that both ends already hang under the same point; the rest is united
// with the lock-free find / compare-and-swap above.  (Before pass 1 existed every edge paid two finds and a compare-and-swap, the lanes
// of a run with k < i one winner at a time: 77 us at N = 16 000; hooking the run's k to the run's minimum instead: 60 us.)
__global__ void cc_union_kernel(const int32_t* __restrict__ edges, unsigned long long ne, const unsigned long long* __restrict__ ne_dev,
                                const int32_t* __restrict__ cnt, int min_samples,
                                int* __restrict__ parent) {
  if (ne_dev) { const unsigned long long d = *ne_dev; ne = d < ne ? d : ne; }
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (unsigned long long)gridDim.x * blockDim.x) {
    const int i = edges[2 * e], k = edges[2 * e + 1];
    if (i == k || cnt[i] < min_samples || cnt[k] < min_samples) continue;
    const int hi = parent[i], hk = parent[k];
    if (hi == hk) continue;
    const int gi = parent[hi], gk = parent[hk];
    if (gi != gk) uf_union(parent, gi, gk);
  }
}
// rootflag[i] = 1 iff i is a core point that is the root (= smallest index) of its component
__global__ void cc_flatten_kernel(const int32_t* __restrict__ cnt, int N, int min_samples, int* __restrict__ parent, int32_t* __restrict__ rootflag) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= N) return;
  int r = i;
  if (cnt[i] >= min_samples) { r = uf_find(parent, i); parent[i] = r; }
  rootflag[i] = (cnt[i] >= min_samples && r == i) ? 1 : 0;
}
__global__ void cc_label_core_kernel(const int32_t* __restrict__ cnt, int N, int min_samples, const int* __restrict__ parent,
                                     const int64_t* __restrict__ rootid, int* __restrict__ lab) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < N && cnt[i] >= min_samples) lab[i] = (int)rootid[parent[i]];
}
// border points take the smallest label among their core neighbours; the same launch labels the core points (disjoint entries of lab)
__global__ void cc_border_kernel(const int32_t* __restrict__ edges, unsigned long long ne, const unsigned long long* __restrict__ ne_dev,
                                 const int32_t* __restrict__ cnt, int min_samples,
                                 const int* __restrict__ parent, const int64_t* __restrict__ rootid, int* __restrict__ lab, int N) {
  if (ne_dev) { const unsigned long long d = *ne_dev; ne = d < ne ? d : ne; }
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = gid; i < (unsigned long long)N; i += stride)
    if (cnt[i] >= min_samples) lab[i] = (int)rootid[parent[i]];
  for (unsigned long long e = gid; e < ne; e += stride) {
    const int i = edges[2 * e], k = edges[2 * e + 1];
    if (cnt[i] >= min_samples && cnt[k] < min_samples) atomicMin(&lab[k], (int)rootid[parent[i]]);
  }
}
__global__ void cc_finalize_kernel(const int* __restrict__ lab, int N, int64_t* __restrict__ labels) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < N) labels[i] = lab[i] == 0x7fffffff ? -1 : (int64_t)lab[i];
}

}  // namespace ssg

using namespace ssg;
// exscan_kernel (exclusive scan that also clears its input) comes from jaccard.hip: the
// library is built as one translation unit (ssg_hip.hip).

static MatView make_view(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value) {
  MatView mv; mv.M = M; mv.v = v; mv.N = N; mv.row0 = row0; mv.nrows = nrows; mv.mode = mode; mv.lambda_value = lambda_value;
  return mv;
}
static int check_view(const char* fn, const void* M, const uint16_t* v, int N, int row0, int nrows, int mode) {
  if (!M || N <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > N || mode < 0 || mode > 2 || (mode == 0 && !v)) {
    ssg_set_error("%s: bad matrix view (N=%d row0=%d nrows=%d mode=%d)", fn, N, row0, nrows, mode);
    return SSG_ERR_INVALID;
  }
  return SSG_OK;
}
// Persistent grid: ~5 workgroups per CU.  Every wave walks many rows and publishes its staged results with one
// cursor atomic per ~500 entries; one wave per row meant N cursor / histogram atomics on a single word
// (one word saturates at ~90 atomics/us: 16 000 rows = 0.18 ms of pure serialisation).
// The grid is then trimmed so that every wave walks the SAME number of rows: 16 000 rows on 5120 waves are 3.1 rows per wave, i.e.
// most waves finish after 3 rows and wait for the ones that got 4 (region query 0.18 ms); 4000 waves x exactly 4 rows: 0.15 ms.
static int stream_grid(int nrows) {
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("SSG_STREAM_GRID"); cap = e ? atoi(e) : 1280; }     // tuning knob
  const int b = (nrows + 3) / 4;                        // one row per wave, four waves per workgroup
  if (b <= cap) return b;
  const int per_wave = (nrows + 4 * cap - 1) / (4 * cap);          // rows per wave at the cap
  return (nrows + 4 * per_wave - 1) / (4 * per_wave);
}

extern "C" int ssg_eps_hist(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value,
                            uint64_t prefix, int shift, int width, int count_nonzero, uint64_t* hist, hipStream_t stream) {
  int rc = check_view("ssg_eps_hist", M, v, N, row0, nrows, mode); if (rc) return rc;
  if (width <= 0 || width > 12 || shift < 0 || shift + width > 64) { ssg_set_error("ssg_eps_hist: bad digit"); return SSG_ERR_INVALID; }
#define SSG_HIST(MD) hipLaunchKernelGGL(eps_hist_kernel<MD>, dim3(stream_grid(nrows)), dim3(256), 0, stream, make_view(M, v, N, row0, nrows, mode, lambda_value), \
                     (unsigned long long)prefix, shift, width, count_nonzero, (unsigned long long*)hist)
  if (mode == 0) SSG_HIST(0); else if (mode == 1) SSG_HIST(1); else SSG_HIST(2);
#undef SSG_HIST
  SSG_LAUNCH_CHECK("eps_hist_kernel");
  return SSG_OK;
}

extern "C" int ssg_eps_compact(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value,
                               uint64_t key_max, uint64_t* buf, uint64_t cap, uint64_t* cursor, hipStream_t stream) {
  int rc = check_view("ssg_eps_compact", M, v, N, row0, nrows, mode); if (rc) return rc;
#define SSG_COMPACT(MD) hipLaunchKernelGGL(eps_compact_kernel<MD>, dim3(stream_grid(nrows)), dim3(256), 0, stream, make_view(M, v, N, row0, nrows, mode, lambda_value), \
                     (unsigned long long)key_max, (unsigned long long*)buf, (unsigned long long)cap, (unsigned long long*)cursor)
  if (mode == 0) SSG_COMPACT(0); else if (mode == 1) SSG_COMPACT(1); else SSG_COMPACT(2);
#undef SSG_COMPACT
  SSG_LAUNCH_CHECK("eps_compact_kernel");
  return SSG_OK;
}


// ---- K10 fast path (see eps_sample_hist_kernel): hist = 4097 uint64 zeroed by the caller; thr3 = 3 uint64 out
extern "C" int ssg_eps_sample_hist(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, int row_stride,
                                   const uint64_t* refine, uint64_t* hist, hipStream_t stream) {
  int rc = check_view("ssg_eps_sample_hist", M, v, N, row0, nrows, mode); if (rc) return rc;
  if (row_stride < 1) { ssg_set_error("ssg_eps_sample_hist: row_stride must be >= 1"); return SSG_ERR_INVALID; }
  const int nsamp = (nrows + row_stride - 1) / row_stride;
#define SSG_SH(MD) hipLaunchKernelGGL(eps_sample_hist_kernel<MD>, dim3(stream_grid(nsamp * SPARTS)), dim3(256), 0, stream, make_view(M, v, N, row0, nrows, mode, lambda_value), \
                     row_stride, (const unsigned long long*)refine, (unsigned long long*)hist)
  if (mode == 0) SSG_SH(0); else if (mode == 1) SSG_SH(1); else SSG_SH(2);
#undef SSG_SH
  SSG_LAUNCH_CHECK("eps_sample_hist_kernel");
  return SSG_OK;
}
// round 6: both sampling levels with their selections in TWO launches instead of four (the workgroup that finishes last selects).
// hist2x = 2 x 4097 words and tickets = 2 words, zeroed by the caller; thr5 receives what ssg_eps_select_threshold + ssg_eps_refine_threshold leave.
extern "C" int ssg_eps_sample_threshold(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, int row_stride,
                                        double quantile, uint64_t* hist2x, uint64_t* thr5, uint32_t* tickets2, uint64_t* splitters1023, hipStream_t stream) {
  int rc = check_view("ssg_eps_sample_threshold", M, v, N, row0, nrows, mode); if (rc) return rc;
  if (row_stride < 1 || !(quantile > 0.0) || !hist2x || !thr5 || !tickets2) { ssg_set_error("ssg_eps_sample_threshold: bad arguments"); return SSG_ERR_INVALID; }
  const int nsamp = (nrows + row_stride - 1) / row_stride;
  const MatView mv = make_view(M, v, N, row0, nrows, mode, lambda_value);
  const unsigned g = (unsigned)stream_grid(nsamp * SPARTS);
#define SSG_ST(MD, LVL) hipLaunchKernelGGL(eps_sample_hist_kernel<MD>, dim3(g), dim3(256), 0, stream, mv, row_stride, (LVL) ? (const unsigned long long*)thr5 : nullptr, \
                     (unsigned long long*)hist2x + (LVL) * 4097, quantile, (unsigned long long*)thr5, (unsigned int*)tickets2 + (LVL), (unsigned long long*)splitters1023)
  for (int lvl = 0; lvl < 2; lvl++) { if (mode == 0) SSG_ST(0, lvl); else if (mode == 1) SSG_ST(1, lvl); else SSG_ST(2, lvl); }
#undef SSG_ST
  SSG_LAUNCH_CHECK("eps_sample_hist_kernel (+ selection)");
  return SSG_OK;
}
extern "C" int ssg_eps_select_threshold(const uint64_t* hist, double quantile, uint64_t* thr3, hipStream_t stream) {
  if (!(quantile > 0.0)) { ssg_set_error("ssg_eps_select_threshold: quantile must be positive"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(eps_select_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)hist, quantile, (unsigned long long*)thr3);
  SSG_LAUNCH_CHECK("eps_select_kernel");
  return SSG_OK;
}
extern "C" int ssg_eps_refine_threshold(const uint64_t* hist2, uint64_t* thr5, hipStream_t stream) {
  hipLaunchKernelGGL(eps_select2_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)hist2, (unsigned long long*)thr5);
  SSG_LAUNCH_CHECK("eps_select2_kernel");
  return SSG_OK;
}
extern "C" int ssg_eps_compact_below(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value,
                                     const uint64_t* thr3, uint64_t* buf, uint64_t cap, uint64_t* cursor2, hipStream_t stream) {
  int rc = check_view("ssg_eps_compact_below", M, v, N, row0, nrows, mode); if (rc) return rc;
#define SSG_CT(MD) hipLaunchKernelGGL(eps_compact_thr_kernel<MD>, dim3(stream_grid(nrows)), dim3(256), 0, stream, make_view(M, v, N, row0, nrows, mode, lambda_value), \
                     (const unsigned long long*)thr3, (unsigned long long*)buf, (unsigned long long)cap, (unsigned long long*)cursor2, (const unsigned long long*)nullptr, (const unsigned char*)nullptr)
  if (mode == 0) SSG_CT(0); else if (mode == 1) SSG_CT(1); else SSG_CT(2);
#undef SSG_CT
  SSG_LAUNCH_CHECK("eps_compact_thr_kernel");
  return SSG_OK;
}

static SparseView make_sparse(const uint32_t* pool, const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                              uint16_t jp0) {
  SparseView sv; sv.pool = pool; sv.seg_off = seg_off; sv.seg_len = seg_len; sv.nseg = nseg; sv.s_cursor = (const unsigned long long*)s_cursor; sv.vmin = vmin; sv.jp0 = jp0;
  return sv;
}

// The same pass through the sparse copy S of J' (ssg_jaccard_rows2): cursor3 = {keys collected, exact zeros, dense pass needed}, zeroed by
// the caller; vmin = half bits of min(v) (ssg_half_min); rowmask = nrows bytes of workspace.  Two launches: the walk over S, which takes
// every row whose floor J'(0) + lambda * half(v_i + vmin) lies at or above the threshold and flags the others (all of them when S
// overflowed or lambda < 0), and the dense pass over the flagged rows, gated on cursor3[2] -- no host decision, no read-back in between.
extern "C" int ssg_eps_compact_below_s(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, const uint64_t* thr3,
                                       uint64_t* buf, uint64_t cap, uint64_t* cursor3, const uint32_t* s_pool, const int64_t* seg_off,
                                       const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin, uint16_t jp0_half,
                                       uint8_t* rowmask, hipStream_t stream) {
  int rc = check_view("ssg_eps_compact_below_s", M, v, N, row0, nrows, 0); if (rc) return rc;
  if (!s_pool || !seg_off || !seg_len || nseg <= 0 || !s_cursor || !vmin || !rowmask) { ssg_set_error("ssg_eps_compact_below_s: no sparse copy"); return SSG_ERR_INVALID; }
  const MatView mv = make_view(M, v, N, row0, nrows, 0, lambda_value);
  hipLaunchKernelGGL(eps_compact_sparse_kernel, dim3(stream_grid(nrows)), dim3(256), 0, stream, mv, make_sparse(s_pool, seg_off, seg_len, nseg, s_cursor, vmin, jp0_half),
                     (const unsigned long long*)thr3, (unsigned long long*)buf, (unsigned long long)cap, (unsigned long long*)cursor3, rowmask);
  hipLaunchKernelGGL(eps_compact_thr_kernel<0>, dim3(stream_grid(nrows)), dim3(256), 0, stream, mv, (const unsigned long long*)thr3, (unsigned long long*)buf,
                     (unsigned long long)cap, (unsigned long long*)cursor3, (const unsigned long long*)(cursor3 + 2), (const unsigned char*)rowmask);
  SSG_LAUNCH_CHECK("eps_compact_sparse_kernel");
  return SSG_OK;
}

namespace ssg {
// smallest of N non-negative halves (bit order == value order), one workgroup
__global__ __launch_bounds__(1024) void half_min_kernel(const hbits* __restrict__ v, int N, uint32_t* __restrict__ out) {
  __shared__ unsigned smin[16];
  unsigned m = 0xffffu;
  for (int i = (int)threadIdx.x; i < N; i += 1024) { const unsigned x = v[i]; m = x < m ? x : m; }
  for (int sh = 1; sh < 64; sh <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, sh, 64); m = o < m ? o : m; }
  if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < 16; w++) m = smin[w] < m ? smin[w] : m; *out = m; }
}
}  // namespace ssg
// *out_bits = min_k v[k] as half bits (v >= 0: the source vector of rerank.py:38-40); the row floors of the sparse passes need it
extern "C" int ssg_half_min(const uint16_t* v, int N, uint32_t* out_bits, hipStream_t stream) {
  if (!v || N <= 0 || !out_bits) { ssg_set_error("ssg_half_min: bad arguments"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(ssg::half_min_kernel, dim3(1), dim3(1024), 0, stream, v, N, out_bits);
  SSG_LAUNCH_CHECK("half_min_kernel");
  return SSG_OK;
}

// ascending sort of buf[0..n) (n = power of two >= 2048; pad with ~0)
extern "C" int ssg_sort_u64(uint64_t* buf, uint64_t n, hipStream_t stream) {
  if (n < (uint64_t)SORT_CH || (n & (n - 1))) { ssg_set_error("ssg_sort_u64: n must be a power of two >= %d", SORT_CH); return SSG_ERR_INVALID; }
  unsigned long long* a = (unsigned long long*)buf;
  hipLaunchKernelGGL(bitonic_local_kernel, dim3((unsigned)(n / SORT_CH)), dim3(1024), 0, stream, a, 2ULL, (unsigned long long)SORT_CH);
  for (unsigned long long k = 2ULL * SORT_CH; k <= n; k <<= 1) {
    unsigned long long j = k >> 1;
    while (j >= (unsigned long long)SORT_CH) {                  // global stages j, j/2, ... SORT_CH: three per launch while they last
      int ns = 0;
      for (unsigned long long q = j; q >= (unsigned long long)SORT_CH && ns < 3; q >>= 1) ns++;
      const unsigned long long thr = n >> ns;
      const unsigned blocks = (unsigned)((thr + 255) / 256);
      if (ns == 3) hipLaunchKernelGGL(bitonic_global_kernel<3>, dim3(blocks), dim3(256), 0, stream, a, (unsigned long long)n, k, j);
      else if (ns == 2) hipLaunchKernelGGL(bitonic_global_kernel<2>, dim3(blocks), dim3(256), 0, stream, a, (unsigned long long)n, k, j);
      else hipLaunchKernelGGL(bitonic_global_kernel<1>, dim3(blocks), dim3(256), 0, stream, a, (unsigned long long)n, k, j);
      j >>= ns;
    }
    hipLaunchKernelGGL(bitonic_local_kernel, dim3((unsigned)(n / SORT_CH)), dim3(1024), 0, stream, a, k, k);
  }
  SSG_LAUNCH_CHECK("bitonic sort");
  return SSG_OK;
}

extern "C" int ssg_fill_u64(uint64_t* buf, uint64_t n0, uint64_t n1, uint64_t value, hipStream_t stream) {
  if (n1 > n0) hipLaunchKernelGGL(fill_u64_kernel, dim3(256), dim3(256), 0, stream, (unsigned long long*)buf, n0, n1, value);
  SSG_LAUNCH_CHECK("fill_u64_kernel");
  return SSG_OK;
}

// ---- round 6: fixed-capacity segments (the all-gathered candidate / edge buffers of the ranks) -> one list, counts left on the device ----
// Segment s holds min(counts[s * count_stride], seg_cap) valid 8-byte items at in + s * seg_stride; they go to out back to back in segment
// order.  total2[0] = the number written, total2[1] = 1 when a segment's count exceeded seg_cap (its owner's buffer overflowed: the caller's
// check must fail).  One workgroup per (segment, 4096-item chunk); every workgroup re-derives its segment's base from the <= 64 counts.
namespace ssg {
__global__ __launch_bounds__(256) void concat_segments_kernel(const unsigned long long* __restrict__ in, int nseg, unsigned long long seg_cap,
                                                              unsigned long long seg_stride, const unsigned long long* __restrict__ counts, int count_stride,
                                                              unsigned long long* __restrict__ out, unsigned long long* __restrict__ total2, int chunks_per_seg) {
  const int s = (int)blockIdx.x / chunks_per_seg, ch = (int)blockIdx.x % chunks_per_seg;
  unsigned long long base = 0, mine = 0, over = 0, sum = 0;
  for (int r = 0; r < nseg; r++) {
    unsigned long long c = counts[(size_t)r * count_stride];
    if (c > seg_cap) { c = seg_cap; over = 1; }
    if (r < s) base += c;
    if (r == s) mine = c;
    sum += c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { total2[0] = sum; total2[1] = over; }
  const unsigned long long i0 = (unsigned long long)ch * 4096;
  const unsigned long long* src = in + (size_t)s * seg_stride;
#pragma unroll 4
  for (unsigned long long i = i0 + threadIdx.x; i < i0 + 4096 && i < mine; i += 256) out[base + i] = src[i];
}
}  // namespace ssg
extern "C" int ssg_concat_segments_u64(const uint64_t* in, int nseg, uint64_t seg_cap, uint64_t seg_stride, const uint64_t* counts, int count_stride,
                                       uint64_t* out, uint64_t* total2, hipStream_t stream) {
  if (!in || !counts || !out || !total2 || nseg <= 0 || nseg > 4096 || seg_stride < seg_cap || count_stride < 1) {
    ssg_set_error("ssg_concat_segments_u64: bad arguments (nseg=%d)", nseg);
    return SSG_ERR_INVALID;
  }
  const int cps = (int)((seg_cap + 4095) / 4096) > 0 ? (int)((seg_cap + 4095) / 4096) : 1;
  hipLaunchKernelGGL(ssg::concat_segments_kernel, dim3((unsigned)(nseg * cps)), dim3(256), 0, stream, (const unsigned long long*)in, nseg, (unsigned long long)seg_cap,
                     (unsigned long long)seg_stride, (const unsigned long long*)counts, count_stride, (unsigned long long*)out, (unsigned long long*)total2, cps);
  SSG_LAUNCH_CHECK("concat_segments_kernel");
  return SSG_OK;
}

// ---- round 5: the eps rule without a host round trip between its full pass and the mean -------------------------------------------
// The number of collected keys stays on the device (cursor[0] of the compaction pass).  The sort network is LAUNCHED for the capacity
// n_cap (a power of two) but every kernel works on n_eff = the power of two >= max(*n_dev, SORT_CH) only: merge levels k > n_eff and
// workgroups past n_eff exit at once (a handful of empty launches of ~2 us instead of a blocking read + a restart of the launch queue).
__device__ __forceinline__ unsigned long long sort_n_eff(const unsigned long long* n_dev, unsigned long long n_cap) {
  unsigned long long n = *n_dev, e = (unsigned long long)SORT_CH;
  if (n > n_cap) n = n_cap;                     // (overflow of the compaction buffer: the check kernel reports it, the result is discarded)
  while (e < n) e <<= 1;
  return e;
}
__global__ void fill_u64_dev_kernel(unsigned long long* p, const unsigned long long* __restrict__ n_dev, unsigned long long n_cap, unsigned long long v) {
  const unsigned long long n1 = sort_n_eff(n_dev, n_cap);
  unsigned long long n0 = *n_dev; if (n0 > n_cap) n0 = n_cap;
  for (unsigned long long i = n0 + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n1; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(1024) void bitonic_local_dev_kernel(unsigned long long* __restrict__ a, unsigned long long k_first, unsigned long long k_last,
                                                                 const unsigned long long* __restrict__ n_dev, unsigned long long n_cap) {
  __shared__ unsigned long long s[SORT_CH];
  const unsigned long long n_eff = sort_n_eff(n_dev, n_cap);
  const unsigned long long g0 = (unsigned long long)blockIdx.x * SORT_CH;
  if (g0 >= n_eff || k_first > n_eff) return;           // (uniform per workgroup)
  const int t = (int)threadIdx.x;
  s[t] = a[g0 + t]; s[t + 1024] = a[g0 + t + 1024];
  __syncthreads();
  for (unsigned long long k = k_first; k <= k_last; k <<= 1) {
    for (unsigned long long j = (k >> 1) < (unsigned long long)(SORT_CH / 2) ? (k >> 1) : (unsigned long long)(SORT_CH / 2); j > 0; j >>= 1) {
      const unsigned long long i = ((unsigned long long)t / j) * (2 * j) + ((unsigned long long)t % j);
      const bool up = (((g0 + i) & k) == 0);
      unsigned long long x = s[i], y = s[i + j];
      cswap(x, y, up);
      s[i] = x; s[i + j] = y;
      __syncthreads();
    }
  }
  a[g0 + t] = s[t]; a[g0 + t + 1024] = s[t + 1024];
}
template <int NS>
__global__ __launch_bounds__(256) void bitonic_global_dev_kernel(unsigned long long* __restrict__ a, unsigned long long k, unsigned long long j,
                                                                 const unsigned long long* __restrict__ n_dev, unsigned long long n_cap) {
  constexpr int NE = 1 << NS;
  const unsigned long long n_eff = sort_n_eff(n_dev, n_cap);
  if (k > n_eff) return;
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_eff / NE) return;
  const unsigned long long jl = j >> (NS - 1);
  const unsigned long long base = (t / jl) * (NE * jl) + (t % jl);
  const bool up = ((base & k) == 0);
  unsigned long long e[NE];
#pragma unroll
  for (int b = 0; b < NE; b++) e[b] = a[base + (unsigned long long)b * jl];
#pragma unroll
  for (int s = NS - 1; s >= 0; s--) {
#pragma unroll
    for (int b = 0; b < NE; b++)
      if (!(b & (1 << s))) cswap(e[b], e[b | (1 << s)], up);
  }
#pragma unroll
  for (int b = 0; b < NE; b++) a[base + (unsigned long long)b * jl] = e[b];
}
// ascending sort of buf[0 .. *n_dev) inside a buffer of n_cap entries (n_cap = power of two >= 2048); entries [*n_dev, n_eff) are
// overwritten with ~0 first.  After the call buf[0 .. *n_dev) is sorted (when *n_dev <= n_cap).
extern "C" int ssg_sort_u64_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, hipStream_t stream) {
  if (n_cap < (uint64_t)SORT_CH || (n_cap & (n_cap - 1)) || !n_dev) { ssg_set_error("ssg_sort_u64_dev: n_cap must be a power of two >= %d, n_dev non-null", SORT_CH); return SSG_ERR_INVALID; }
  unsigned long long* a = (unsigned long long*)buf;
  const unsigned long long* nd = (const unsigned long long*)n_dev;
  const unsigned long long n = n_cap;
  hipLaunchKernelGGL(fill_u64_dev_kernel, dim3(256), dim3(256), 0, stream, a, nd, n, ~0ULL);
  hipLaunchKernelGGL(bitonic_local_dev_kernel, dim3((unsigned)(n / SORT_CH)), dim3(1024), 0, stream, a, 2ULL, (unsigned long long)SORT_CH, nd, n);
  for (unsigned long long k = 2ULL * SORT_CH; k <= n; k <<= 1) {
    unsigned long long j = k >> 1;
    while (j >= (unsigned long long)SORT_CH) {
      int ns = 0;
      for (unsigned long long q = j; q >= (unsigned long long)SORT_CH && ns < 3; q >>= 1) ns++;
      const unsigned long long thr = n >> ns;
      const unsigned blocks = (unsigned)((thr + 255) / 256);
      if (ns == 3) hipLaunchKernelGGL(bitonic_global_dev_kernel<3>, dim3(blocks), dim3(256), 0, stream, a, k, j, nd, n);
      else if (ns == 2) hipLaunchKernelGGL(bitonic_global_dev_kernel<2>, dim3(blocks), dim3(256), 0, stream, a, k, j, nd, n);
      else hipLaunchKernelGGL(bitonic_global_dev_kernel<1>, dim3(blocks), dim3(256), 0, stream, a, k, j, nd, n);
      j >>= ns;
    }
    hipLaunchKernelGGL(bitonic_local_dev_kernel, dim3((unsigned)(n / SORT_CH)), dim3(1024), 0, stream, a, k, k, nd, n);
  }
  SSG_LAUNCH_CHECK("bitonic sort (device-sized)");
  return SSG_OK;
}

// ---- round 5: sample sort of the device-sized key list -------------------------------------------------------------------------------
// The bitonic network above costs 25 launches (0.15 ms of 4..9 us kernels + their dispatch gaps) for the ~2.7e5 candidates of the eps rule
// at N = 16 000 -- a third of the whole eps rule + DBSCAN chain.  Five launches instead:
//   1. one workgroup sorts a strided sample of 4096 keys in LDS and keeps every 4th as a splitter (1023 of them);
//   2. every workgroup of 4096 keys finds each key's bucket by a 10-step lower bound over the splitters in LDS and counts per bucket:
//      bucket = 2 * #(splitters < key) + (key == that splitter) -- keys EQUAL to a splitter get a bucket of their own (odd numbers), so
//      any number of duplicates (half-valued distances: a few hundred distinct keys) stays out of the buckets that need sorting; the
//      workgroup reserves its share of every bucket with one returning atomic per non-empty bucket (the order inside a bucket is free);
//   3. one workgroup scans the 2048 bucket totals into offsets;
//   4. the keys are scattered to their bucket's range (LDS cursors on top of the reserved bases);
//   5. one workgroup per bucket sorts its keys in LDS (bitonic on the next power of two: ~260 keys at 4x oversampling) and writes them
//      back.  Buckets over 2048 keys are ranked out of global memory (slow, exact); over 16384: *fail = 1.
// (First version, measured: ranking every key against the whole bucket in LDS = 73 us of dependent LDS reads, the sample's bitonic
//  network with integer divisions in its index arithmetic = 51 us, a (workgroup x bucket) count matrix scanned by one workgroup = 17 us:
//  171 us, slower than the network it was to replace.)
// Two geometries (round 6): LB = 10 -- 1023 splitters out of a sorted sample of 4096 keys, chunks of 4096 keys per counting workgroup,
// sorting buckets of up to 2048 keys (256 threads) -- while the expected number of keys stays below ~4e5 (LDS-sized buckets with a 4x
// margin); LB = 12 -- 4095 splitters out of a sample of 16 384, chunks of 16 384, buckets of up to 16 384 keys in 128 KB of LDS (1024
// threads, up to 16 keys per thread in registers) -- up to ~2.4e7 keys (N = 128 000: 1.7e7 candidates, 4.1 k per bucket on average), where
// the bitonic network over the whole array took 7.1 ms.
template <int LB> struct SsCfg {
  static constexpr int NSLOT = 1 << LB, NSPLIT = NSLOT - 1, NBK = 2 * NSLOT, SAMPLE = 4 * NSLOT, CHUNK = LB == 10 ? 4096 : 16384;
  static constexpr int CAP = LB == 10 ? 2048 : 16384, NTB = LB == 10 ? 256 : 1024;     // largest bucket sorted in LDS, threads of the bucket sort
  static constexpr int CAP_GLOBAL = 16384;                                              // LB = 10: ranked out of global memory up to here; beyond: fail
};
struct SsWs { unsigned long long* tmp; unsigned long long* split; unsigned int* cmat; unsigned int* gcount; unsigned int* off; unsigned short* bid; };
template <int LB>
static size_t ss_layout(uint64_t n_cap, char* base, SsWs* w) {
  using C = SsCfg<LB>;
  size_t o = 0;
  const size_t G = (size_t)((n_cap + C::CHUNK - 1) / C::CHUNK);
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  const size_t a_tmp = take((size_t)n_cap * 8), a_split = take((size_t)C::NSLOT * 8), a_cmat = take(G * C::NBK * 4), a_gc = take((size_t)C::NBK * 4),
               a_off = take(((size_t)C::NBK + 1) * 4), a_bid = take((size_t)n_cap * 2);
  if (w) {
    w->tmp = (unsigned long long*)(base + a_tmp); w->split = (unsigned long long*)(base + a_split); w->cmat = (unsigned int*)(base + a_cmat);
    w->gcount = (unsigned int*)(base + a_gc); w->off = (unsigned int*)(base + a_off); w->bid = (unsigned short*)(base + a_bid);
  }
  return o;
}
__device__ __forceinline__ unsigned long long ss_count_of(const unsigned long long* n_dev, unsigned long long n_cap) {
  const unsigned long long n = *n_dev;
  return n > n_cap ? n_cap : n;
}
// One bitonic network over M = E * NT keys, E consecutive keys per thread IN REGISTERS (key index e = t * E + r).  A compare-exchange
// partner at distance j sits in the same thread (j < E), in the same wave (j < 64 E: one 64-bit shuffle) or in another wave (LDS, two
// barriers) -- 10 of the 78 steps of a 4096-key network go through LDS instead of all of them (the all-LDS version was bound by LDS
// traffic: 40 us for the sample, 21 us for the buckets).  After the call key e of the sorted order is v[e - t * E] of thread e / E.
template <int E, int NT>
__device__ __forceinline__ void ss_bitonic_reg(unsigned long long (&v)[E], unsigned long long* s, int t) {
  constexpr int M = E * NT, WSPAN = 64 * E;
  const int e0 = t * E;
  for (int k = 2; k <= M; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int jj = 1; jj < E; jj <<= 1)
          if (j == jj) {
#pragma unroll
            for (int r = 0; r < E; r++)
              if (!(r & jj)) cswap(v[r], v[r | jj], ((e0 + r) & k) == 0);
          }
      } else {
        unsigned long long pv[E];
        if (j < WSPAN) {
          const int lx = j / E;
#pragma unroll
          for (int r = 0; r < E; r++) pv[r] = __shfl_xor(v[r], lx, 64);
        } else {
#pragma unroll
          for (int r = 0; r < E; r++) s[e0 + r] = v[r];
          __syncthreads();
#pragma unroll
          for (int r = 0; r < E; r++) pv[r] = s[(e0 + r) ^ j];
          __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < E; r++) {
          const int e = e0 + r;
          const bool take_min = (((e & j) == 0) == ((e & k) == 0));      // the lower key of an ascending pair, the upper one of a descending pair
          const unsigned long long a = v[r], b = pv[r];
          v[r] = take_min ? (a < b ? a : b) : (a < b ? b : a);
        }
      }
    }
}
template <int LB>
__global__ __launch_bounds__(1024) void ss_splitters_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ n_dev, unsigned long long n_cap,
                                                            unsigned long long* __restrict__ split, unsigned int* __restrict__ gcount,
                                                            unsigned long long* __restrict__ fail) {
  using C = SsCfg<LB>;
  constexpr int E = C::SAMPLE / 1024;
  __shared__ unsigned long long s[C::SAMPLE];
  const unsigned long long n = ss_count_of(n_dev, n_cap);
  const int t = (int)threadIdx.x;
  if (t == 0) *fail = 0ull;
  for (int q = t; q < C::NBK; q += 1024) gcount[q] = 0u;
  if (n == 0) return;
  unsigned long long v[E];
#pragma unroll
  for (int r = 0; r < E; r++) v[r] = a[((unsigned long long)(E * t + r) * n) >> (LB + 2)];        // (n < 2^31, sample index < 2^14)
  ss_bitonic_reg<E, 1024>(v, s, t);
  // sorted sample keys 3, 7, 11, ...: with E keys per thread the splitters of thread t are its keys 3, 7, ... (E = 4: one, E = 16: four)
#pragma unroll
  for (int r = 3; r < E; r += 4) { const int q = (E * t + r) >> 2; if (q < C::NSPLIT) split[q] = v[r]; }
}
// sp: NSLOT entries in LDS, the NSPLIT splitters ascending + ~0
template <int NSLOT>
__device__ __forceinline__ int ss_bucket(const unsigned long long* sp, unsigned long long key) {
  int lo = 0;
#pragma unroll
  for (int step = NSLOT / 2; step > 0; step >>= 1) lo += (sp[lo + step - 1] < key) ? step : 0;
  return 2 * lo + (sp[lo] == key ? 1 : 0);
}
template <int LB>
__global__ __launch_bounds__(256) void ss_count_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ n_dev, unsigned long long n_cap,
                                                       const unsigned long long* __restrict__ split, unsigned int* __restrict__ gcount,
                                                       unsigned int* __restrict__ cmat, unsigned short* __restrict__ bid) {
  using C = SsCfg<LB>;
  __shared__ unsigned long long sp[C::NSLOT];
  __shared__ unsigned int cn[C::NBK];
  const unsigned long long n = ss_count_of(n_dev, n_cap);
  const unsigned long long g0 = (unsigned long long)blockIdx.x * C::CHUNK;
  if (g0 >= n) return;
  const int t = (int)threadIdx.x;
  for (int q = t; q < C::NSLOT; q += 256) sp[q] = q < C::NSPLIT ? split[q] : ~0ull;
  for (int q = t; q < C::NBK; q += 256) cn[q] = 0u;
  __syncthreads();
  for (int u = 0; u < C::CHUNK / 256; u += 4) {
    unsigned long long key[4]; int b[4];
#pragma unroll
    for (int v = 0; v < 4; v++) { const unsigned long long idx = g0 + (unsigned long long)(u + v) * 256 + t; key[v] = a[idx < n ? idx : n - 1]; }
#pragma unroll
    for (int v = 0; v < 4; v++) b[v] = ss_bucket<C::NSLOT>(sp, key[v]);
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const unsigned long long idx = g0 + (unsigned long long)(u + v) * 256 + t;
      if (idx < n) { atomicAdd(&cn[b[v]], 1u); bid[idx] = (unsigned short)b[v]; }
    }
  }
  __syncthreads();
  for (int q = t; q < C::NBK; q += 256) {
    const unsigned int c = cn[q];
    cmat[(size_t)blockIdx.x * C::NBK + q] = c ? atomicAdd(&gcount[q], c) : 0u;       // this workgroup's base inside bucket q
  }
}
// (round 6: every workgroup scans the bucket counts itself -- NBK / 256 per thread + a 256-entry LDS scan, ~1 us -- instead of waiting for a
//  one-workgroup scan launch in between; workgroup 0 also leaves the offsets in `off` for the bucket sort)
template <int LB>
__global__ __launch_bounds__(256) void ss_scatter_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ n_dev, unsigned long long n_cap,
                                                         const unsigned int* __restrict__ cmat, const unsigned int* __restrict__ gcount, unsigned int* __restrict__ off,
                                                         const unsigned short* __restrict__ bid, unsigned long long* __restrict__ tmp) {
  using C = SsCfg<LB>;
  constexpr int PT = C::NBK / 256;                      // bucket counts per thread: 8 or 32
  __shared__ unsigned int cur[C::NBK];
  __shared__ unsigned int psum[256];
  const unsigned long long n = ss_count_of(n_dev, n_cap);
  const unsigned long long g0 = (unsigned long long)blockIdx.x * C::CHUNK;
  if (g0 >= n && blockIdx.x != 0) return;
  const int t = (int)threadIdx.x;
  unsigned int mine = 0;
  for (int u = 0; u < PT; u++) mine += gcount[PT * t + u];
  psum[t] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned int add = t >= o ? psum[t - o] : 0u;
    __syncthreads();
    psum[t] += add;
    __syncthreads();
  }
  unsigned int run = psum[t] - mine;
  for (int u = 0; u < PT; u++) {
    const unsigned int gc = gcount[PT * t + u];
    cur[PT * t + u] = run + (g0 < n ? cmat[(size_t)blockIdx.x * C::NBK + PT * t + u] : 0u);
    if (blockIdx.x == 0) off[PT * t + u] = run;
    run += gc;
  }
  if (blockIdx.x == 0 && t == 255) off[C::NBK] = run;
  if (g0 >= n) return;                      // (workgroup 0 of an empty input: only the offsets)
  __syncthreads();
  for (int u = 0; u < C::CHUNK / 256; u += 4) {
    unsigned long long key[4]; int b[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const unsigned long long idx = g0 + (unsigned long long)(u + v) * 256 + t, ic = idx < n ? idx : n - 1;
      key[v] = a[ic]; b[v] = (int)bid[ic];
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const unsigned long long idx = g0 + (unsigned long long)(u + v) * 256 + t;
      if (idx < n) tmp[atomicAdd(&cur[b[v]], 1u)] = key[v];
    }
  }
}
template <int LB>
__global__ __launch_bounds__(SsCfg<LB>::NTB) void ss_bucket_sort_kernel(const unsigned long long* __restrict__ tmp, const unsigned int* __restrict__ off,
                                                                        unsigned long long* __restrict__ out, unsigned long long* __restrict__ fail) {
  using C = SsCfg<LB>;
  constexpr int NTB = C::NTB;
  __shared__ unsigned long long s[C::CAP];
  const int b = (int)blockIdx.x, t = (int)threadIdx.x;
  const unsigned int s0 = off[b];
  const int len = (int)(off[b + 1] - s0);
  if (len <= 0) return;
  const bool hopeless = LB == 10 ? len > C::CAP_GLOBAL : len > C::CAP;
  if ((b & 1) || len == 1 || hopeless) {                             // all keys equal (a splitter's own bucket) / one key / hopeless: copy
    for (int q = t; q < len; q += NTB) out[s0 + q] = tmp[s0 + q];
    if (!(b & 1) && hopeless && t == 0) atomicOr(fail, 1ull);
    return;
  }
  if (len <= C::CAP) {
    // NTB * E keys (E = the power of two that holds the bucket), padded with ~0, the network in registers
#define SS_BUCKET(E_)                                                                                          \
    {                                                                                                          \
      unsigned long long v[E_];                                                                                \
      _Pragma("unroll") for (int r = 0; r < E_; r++) { const int e = t * E_ + r; v[r] = e < len ? tmp[s0 + e] : ~0ull; } \
      ss_bitonic_reg<E_, NTB>(v, s, t);                                                                        \
      _Pragma("unroll") for (int r = 0; r < E_; r++) { const int e = t * E_ + r; if (e < len) out[s0 + e] = v[r]; }      \
    }
    if (len <= NTB) SS_BUCKET(1) else if (len <= 2 * NTB) SS_BUCKET(2) else if (len <= 4 * NTB) SS_BUCKET(4) else if (len <= 8 * NTB) SS_BUCKET(8)
    else if constexpr (C::CAP >= 16 * NTB) SS_BUCKET(16)
#undef SS_BUCKET
    return;
  }
  for (int q = t; q < len; q += NTB) {                                // (LB = 10) an oversized bucket: every key ranked against the bucket out of global memory
    const unsigned long long key = tmp[s0 + q];
    int rank = 0;
    for (int j = 0; j < len; j++) { const unsigned long long kj = tmp[s0 + j]; rank += (kj < key || (kj == key && j < q)) ? 1 : 0; }
    out[s0 + rank] = key;
  }
}
template <int LB>
static int samplesort_impl(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, const uint64_t* splitters, uint32_t* gcount_ext, void* ws, size_t ws_bytes,
                           uint64_t* fail, hipStream_t stream) {
  using C = SsCfg<LB>;
  SsWs w;
  if (ss_layout<LB>(n_cap, (char*)ws, &w) > ws_bytes) { ssg_set_error("sample sort: workspace too small"); return SSG_ERR_INVALID; }
  unsigned long long* a = (unsigned long long*)buf;
  const unsigned long long* nd = (const unsigned long long*)n_dev;
  const unsigned G = (unsigned)((n_cap + C::CHUNK - 1) / C::CHUNK);
  const unsigned long long* sp = splitters ? (const unsigned long long*)splitters : w.split;
  unsigned int* gc = gcount_ext ? gcount_ext : w.gcount;
  if (!splitters) hipLaunchKernelGGL(ss_splitters_kernel<LB>, dim3(1), dim3(1024), 0, stream, a, nd, (unsigned long long)n_cap, w.split, w.gcount, (unsigned long long*)fail);
  hipLaunchKernelGGL(ss_count_kernel<LB>, dim3(G), dim3(256), 0, stream, a, nd, (unsigned long long)n_cap, sp, gc, w.cmat, w.bid);
  hipLaunchKernelGGL(ss_scatter_kernel<LB>, dim3(G), dim3(256), 0, stream, a, nd, (unsigned long long)n_cap, w.cmat, (const unsigned int*)gc, w.off, w.bid, w.tmp);
  hipLaunchKernelGGL(ss_bucket_sort_kernel<LB>, dim3(C::NBK), dim3(C::NTB), 0, stream, w.tmp, w.off, a, (unsigned long long*)fail);
  SSG_LAUNCH_CHECK("sample sort (device-sized)");
  return SSG_OK;
}
extern "C" size_t ssg_samplesort_u64_workspace_bytes(uint64_t n_cap) { return ss_layout<10>(n_cap, nullptr, nullptr); }
// ascending sort of buf[0 .. min(*n_dev, n_cap)) in place (through ws); *fail = 1 when a bucket could not be sorted (more than 16384 keys
// strictly between two neighbouring splitters: the sample missed the distribution) -- buf then holds a permutation of the keys, not sorted.
extern "C" int ssg_samplesort_u64_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, void* ws, size_t ws_bytes, uint64_t* fail, hipStream_t stream) {
  if (!buf || !n_dev || !ws || !fail || n_cap == 0 || n_cap > (1ull << 31)) { ssg_set_error("ssg_samplesort_u64_dev: bad arguments"); return SSG_ERR_INVALID; }
  return samplesort_impl<10>(buf, n_cap, n_dev, nullptr, nullptr, ws, ws_bytes, fail, stream);
}
// round 6: the LB = 12 geometry (see above) for 4e5 .. 2.4e7 expected keys; same contract, its own workspace size
extern "C" size_t ssg_samplesort_u64_big_workspace_bytes(uint64_t n_cap) { return ss_layout<12>(n_cap, nullptr, nullptr); }
extern "C" int ssg_samplesort_u64_big_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, void* ws, size_t ws_bytes, uint64_t* fail, hipStream_t stream) {
  if (!buf || !n_dev || !ws || !fail || n_cap == 0 || n_cap > (1ull << 31)) { ssg_set_error("ssg_samplesort_u64_big_dev: bad arguments"); return SSG_ERR_INVALID; }
  return samplesort_impl<12>(buf, n_cap, n_dev, nullptr, nullptr, ws, ws_bytes, fail, stream);
}
// round 6: the LB = 10 sort on splitters the caller already has on the device (ssg_eps_sample_threshold derives them from its histogram):
// three launches, no sample.  gcount2048 (2048 uint32) and *fail must be ZERO on entry (the caller's one zero-filled allocation).
extern "C" int ssg_samplesort_u64_presplit_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, const uint64_t* splitters1023, uint32_t* gcount2048,
                                               void* ws, size_t ws_bytes, uint64_t* fail, hipStream_t stream) {
  if (!buf || !n_dev || !ws || !fail || !splitters1023 || !gcount2048 || n_cap == 0 || n_cap > (1ull << 31)) { ssg_set_error("ssg_samplesort_u64_presplit_dev: bad arguments"); return SSG_ERR_INVALID; }
  return samplesort_impl<10>(buf, n_cap, n_dev, splitters1023, gcount2048, ws, ws_bytes, fail, stream);
}

// The a-posteriori checks of the sampled eps rule (ssg_amd/cluster.py _eps_rule_sampled) on the device, one thread:
//   count = upper_total - zeros;  top = rint(rho * count) (np.round: half to even, selftraining.py:292) must equal the top_guess the
//   summation tree was built for;  every collected key fits (got <= n_cap), at least `top` were collected, the threshold is finite and
//   the top-th sorted key lies below it by the float32 surrogate's margin (=> the `top` smallest are all among the collected keys);
//   *sort_fail (the sample sort's word; may be null) is zero.
// status6 = {ok, got, zeros, top, bits of the top-th key, threshold bits};  eps2[0] (the mean ssg_eps_mean_run left there) is replaced
// by NaN when a check fails, so that a region query queued behind finds nothing and the caller falls back.
__global__ void eps_check_kernel(const unsigned long long* __restrict__ sorted, const unsigned long long* __restrict__ cursor, const unsigned long long* __restrict__ thr3,
                                 double rho, unsigned long long upper_total, long long top_guess, unsigned long long n_cap, double* __restrict__ eps2,
                                 unsigned long long* __restrict__ status6, const unsigned long long* __restrict__ sort_fail) {
  if (blockIdx.x || threadIdx.x) return;
  EpsCheckArgs a; a.cursor = cursor; a.thr3 = thr3; a.rho = rho; a.upper_total = upper_total; a.top_guess = top_guess; a.n_cap = n_cap; a.status6 = status6; a.sort_fail = sort_fail;
  eps_check_one(sorted, a, eps2);
}
extern "C" int ssg_eps_check(const uint64_t* sorted_keys, const uint64_t* cursor, const uint64_t* thr3, double rho, uint64_t upper_total, int64_t top_guess,
                             uint64_t n_cap, double* eps2, uint64_t* status6, const uint64_t* sort_fail, hipStream_t stream) {
  if (!sorted_keys || !cursor || !thr3 || !eps2 || !status6) { ssg_set_error("ssg_eps_check: null argument"); return SSG_ERR_INVALID; }
  hipLaunchKernelGGL(eps_check_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)sorted_keys, (const unsigned long long*)cursor,
                     (const unsigned long long*)thr3, rho, (unsigned long long)upper_total, (long long)top_guess, (unsigned long long)n_cap, eps2,
                     (unsigned long long*)status6, (const unsigned long long*)sort_fail);
  SSG_LAUNCH_CHECK("eps_check_kernel");
  return SSG_OK;
}

// workspace bytes for ssg_eps_mean: recursion tables + node values for `top` summands
extern "C" size_t ssg_eps_mean_workspace_bytes(int64_t top) { const size_t nl = (size_t)(top / 32 + 64); return nl * 64 + 1024; }

namespace {
struct PwTree {
  std::vector<long long> leaf_off; std::vector<int> leaf_n, node_l, node_r, node_h, level_ptr;
  // returns node id (leaves: index; internal: nleaves + index) -- internal ids are remapped after sorting by height
  int build(long long off, long long n, std::vector<int>& il, std::vector<int>& ir, std::vector<int>& ih, int& height) {
    if (n <= 128) { leaf_off.push_back(off); leaf_n.push_back((int)n); height = 0; return -(int)leaf_off.size(); }   // leaf id = -(idx+1)
    long long n2 = n / 2; n2 -= n2 % 8;
    int hl, hr;
    const int l = build(off, n2, il, ir, ih, hl), r = build(off + n2, n - n2, il, ir, ih, hr);
    height = (hl > hr ? hl : hr) + 1;
    il.push_back(l); ir.push_back(r); ih.push_back(height);
    return (int)il.size() - 1;
  }
};
}  // namespace

namespace {
struct PwLayout { int L, I, hroot; size_t o_off, o_n, o_l, o_r, o_lp, o_val, need; std::vector<char> host; };
// numpy's pairwise-summation tree for `top` summands (depends on top only) and its packed device layout
static void pw_layout(int64_t top, PwLayout& out, bool with_tables) {
  PwTree t; std::vector<int> il, ir, ih; int hroot = 0;
  t.build(0, top, il, ir, ih, hroot);
  const int L = (int)t.leaf_off.size(), I = (int)il.size();
  out.L = L; out.I = I; out.hroot = hroot;
  // pack tables: leaf_off[L] i64 | leaf_n[L] | node_l[I] | node_r[I] | level_ptr[hroot+1] | val[(L+I)] (8 B each)
  out.o_off = 0; out.o_n = out.o_off + (size_t)L * 8; out.o_l = out.o_n + (size_t)L * 4; out.o_r = out.o_l + (size_t)I * 4; out.o_lp = out.o_r + (size_t)I * 4;
  out.o_val = out.o_lp + (size_t)(hroot + 1) * 4; out.o_val = (out.o_val + 15) & ~(size_t)15;
  out.need = out.o_val + (size_t)(L + I) * 8;
  if (!with_tables) return;
  // order internal nodes by height (stable), remap child references
  std::vector<int> order(I), pos(I);
  for (int i = 0; i < I; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ih[a] < ih[b]; });
  for (int i = 0; i < I; i++) pos[order[i]] = i;
  t.node_l.resize(I); t.node_r.resize(I); t.level_ptr.assign(hroot + 1, 0);
  auto ref = [&](int c) { return c < 0 ? (-c - 1) : (L + pos[c]); };
  for (int i = 0; i < I; i++) { const int o = order[i]; t.node_l[i] = ref(il[o]); t.node_r[i] = ref(ir[o]); t.level_ptr[ih[o]]++; }
  // level_ptr[h] currently counts nodes of height h (h>=1) at index h; turn into prefix offsets over levels 1..hroot
  std::vector<int> lp(hroot + 1, 0);
  for (int h = 1; h <= hroot; h++) lp[h] = lp[h - 1] + t.level_ptr[h];
  out.host.assign(out.o_val, 0);
  memcpy(out.host.data() + out.o_off, t.leaf_off.data(), (size_t)L * 8);
  memcpy(out.host.data() + out.o_n, t.leaf_n.data(), (size_t)L * 4);
  if (I) { memcpy(out.host.data() + out.o_l, t.node_l.data(), (size_t)I * 4); memcpy(out.host.data() + out.o_r, t.node_r.data(), (size_t)I * 4); }
  memcpy(out.host.data() + out.o_lp, lp.data(), (size_t)(hroot + 1) * 4);
}
}  // namespace

// Step 1 of ssg_eps_mean: upload the recursion tables for `top` into ws.  Blocks until the copy is done (the tables live in a
// local host buffer) -- call it while the stream is idle (right after `top` became known), not behind the sort.
extern "C" int ssg_eps_mean_prepare(int64_t top, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (top <= 0 || ws_bytes < ssg_eps_mean_workspace_bytes(top)) { ssg_set_error("ssg_eps_mean: top=%lld ws too small", (long long)top); return SSG_ERR_INVALID; }
  PwLayout lay; pw_layout(top, lay, true);
  if (lay.need > ws_bytes) { ssg_set_error("ssg_eps_mean: workspace %zu < %zu", ws_bytes, lay.need); return SSG_ERR_INVALID; }
  SSG_HIP(hipMemcpyAsync(ws, lay.host.data(), lay.o_val, hipMemcpyHostToDevice, stream));
  SSG_HIP(hipStreamSynchronize(stream));   // `lay.host` is a local buffer
  return SSG_OK;
}
// Step 2: the summation itself (asynchronous): leaves in parallel (8 lanes per leaf), then one workgroup walks the tree levels
static int eps_mean_run_impl(const uint64_t* sorted_keys, int64_t top, int mode, void* ws, size_t ws_bytes, double* out2, const EpsCheckArgs& chk, hipStream_t stream) {
  if (top <= 0 || ws_bytes < ssg_eps_mean_workspace_bytes(top)) { ssg_set_error("ssg_eps_mean: top=%lld ws too small", (long long)top); return SSG_ERR_INVALID; }
  PwLayout lay; pw_layout(top, lay, false);
  char* w = (char*)ws;
  const int L = lay.L, hroot = lay.hroot;
  const int lb = (L * 8 + 255) / 256;
  if (mode == 0) {
    hipLaunchKernelGGL(eps_leaf_kernel<double>, dim3(lb), dim3(256), 0, stream, (const unsigned long long*)sorted_keys, L, (const long long*)(w + lay.o_off),
                       (const int*)(w + lay.o_n), (double*)(w + lay.o_val));
    hipLaunchKernelGGL(eps_mean_kernel<double>, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)sorted_keys, (long long)top, L, hroot,
                       (const long long*)(w + lay.o_off), (const int*)(w + lay.o_n), (const int*)(w + lay.o_l), (const int*)(w + lay.o_r),
                       (const int*)(w + lay.o_lp), (double*)(w + lay.o_val), out2, chk);
  } else {
    hipLaunchKernelGGL(eps_leaf_kernel<float>, dim3(lb), dim3(256), 0, stream, (const unsigned long long*)sorted_keys, L, (const long long*)(w + lay.o_off),
                       (const int*)(w + lay.o_n), (float*)(w + lay.o_val));
    hipLaunchKernelGGL(eps_mean_kernel<float>, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)sorted_keys, (long long)top, L, hroot,
                       (const long long*)(w + lay.o_off), (const int*)(w + lay.o_n), (const int*)(w + lay.o_l), (const int*)(w + lay.o_r),
                       (const int*)(w + lay.o_lp), (float*)(w + lay.o_val), out2, chk);
  }
  SSG_LAUNCH_CHECK("eps_mean kernels");
  return SSG_OK;
}
extern "C" int ssg_eps_mean_run(const uint64_t* sorted_keys, int64_t top, int mode, void* ws, size_t ws_bytes, double* out2, hipStream_t stream) {
  return eps_mean_run_impl(sorted_keys, top, mode, ws, ws_bytes, out2, EpsCheckArgs(), stream);
}
// round 6: ssg_eps_mean_run + ssg_eps_check in the launches of the former (the tree kernel's last thread runs the checks)
extern "C" int ssg_eps_mean_check(const uint64_t* sorted_keys, int64_t top_guess, int mode, void* ws, size_t ws_bytes, double* eps2, const uint64_t* cursor,
                                  const uint64_t* thr3, double rho, uint64_t upper_total, uint64_t n_cap, uint64_t* status6, const uint64_t* sort_fail, hipStream_t stream) {
  if (!cursor || !thr3 || !status6 || !eps2) { ssg_set_error("ssg_eps_mean_check: null argument"); return SSG_ERR_INVALID; }
  EpsCheckArgs a; a.cursor = (const unsigned long long*)cursor; a.thr3 = (const unsigned long long*)thr3; a.rho = rho; a.upper_total = upper_total;
  a.top_guess = top_guess; a.n_cap = n_cap; a.status6 = (unsigned long long*)status6; a.sort_fail = (const unsigned long long*)sort_fail;
  return eps_mean_run_impl(sorted_keys, top_guess, mode, ws, ws_bytes, eps2, a, stream);
}
extern "C" int ssg_eps_mean(const uint64_t* sorted_keys, int64_t top, int mode, void* ws, size_t ws_bytes, double* out2, hipStream_t stream) {
  const int rc = ssg_eps_mean_prepare(top, ws, ws_bytes, stream);
  return rc ? rc : ssg_eps_mean_run(sorted_keys, top, mode, ws, ws_bytes, out2, stream);
}

static int region_query_impl(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, double eps, const double* eps_dev,
                             int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor, hipStream_t stream) {
  int rc = check_view("ssg_region_query", M, v, N, row0, nrows, mode); if (rc) return rc;
#define SSG_RQ(MD) hipLaunchKernelGGL(region_query_kernel<MD>, dim3(stream_grid(nrows)), dim3(256), 0, stream, make_view(M, v, N, row0, nrows, mode, lambda_value), eps, \
                     cnt, edges, (unsigned long long)cap_edges, (unsigned long long*)cursor, (const unsigned long long*)nullptr, (const unsigned char*)nullptr, eps_dev)
  if (mode == 0) SSG_RQ(0); else if (mode == 1) SSG_RQ(1); else SSG_RQ(2);
#undef SSG_RQ
  SSG_LAUNCH_CHECK("region_query_kernel");
  return SSG_OK;
}
extern "C" int ssg_region_query(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, double eps,
                                int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor, hipStream_t stream) {
  return region_query_impl(M, v, N, row0, nrows, mode, lambda_value, eps, nullptr, cnt, edges, cap_edges, cursor, stream);
}
// round 5: eps is read from device memory (what ssg_eps_check left there), so that eps rule -> region query -> components run back
// to back without the host seeing eps in between; a NaN there means "the eps rule's fast path did not verify": no hit, the caller redoes
extern "C" int ssg_region_query_dev(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, const double* eps_dev,
                                    int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor, hipStream_t stream) {
  if (!eps_dev) { ssg_set_error("ssg_region_query_dev: null eps"); return SSG_ERR_INVALID; }
  return region_query_impl(M, v, N, row0, nrows, mode, lambda_value, 0.0, eps_dev, cnt, edges, cap_edges, cursor, stream);
}

// Region query through the sparse copy S of J': cursor2 = {edges counted, dense pass needed} zeroed by the caller, rowmask = nrows bytes
// of workspace.  Row i goes through S when J'(0) + lambda * half(v_i + vmin) > eps; the other rows (all of them when S overflowed) are
// flagged and done by the dense pass queued behind, gated on cursor2[1].  Same outputs as ssg_region_query.
static int region_query_s_impl(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double eps, const double* eps_dev,
                               const uint32_t* s_pool, const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                               uint16_t jp0_half, uint8_t* rowmask, int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor2, hipStream_t stream) {
  int rc = check_view("ssg_region_query_s", M, v, N, row0, nrows, 0); if (rc) return rc;
  if (!s_pool || !seg_off || !seg_len || nseg <= 0 || !s_cursor || !vmin || !rowmask) { ssg_set_error("ssg_region_query_s: no sparse copy"); return SSG_ERR_INVALID; }
  const MatView mv = make_view(M, v, N, row0, nrows, 0, lambda_value);
  hipLaunchKernelGGL(region_query_sparse_kernel, dim3(stream_grid(nrows)), dim3(256), 0, stream, mv, make_sparse(s_pool, seg_off, seg_len, nseg, s_cursor, vmin, jp0_half),
                     eps, cnt, edges, (unsigned long long)cap_edges, (unsigned long long*)cursor2, rowmask, eps_dev);
  hipLaunchKernelGGL(region_query_kernel<0>, dim3(stream_grid(nrows)), dim3(256), 0, stream, mv, eps, cnt, edges, (unsigned long long)cap_edges,
                     (unsigned long long*)cursor2, (const unsigned long long*)(cursor2 + 1), (const unsigned char*)rowmask, eps_dev);
  SSG_LAUNCH_CHECK("region_query_sparse_kernel");
  return SSG_OK;
}
extern "C" int ssg_region_query_s(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double eps, const uint32_t* s_pool,
                                  const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                                  uint16_t jp0_half, uint8_t* rowmask, int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor2,
                                  hipStream_t stream) {
  return region_query_s_impl(M, v, N, row0, nrows, lambda_value, eps, nullptr, s_pool, seg_off, seg_len, nseg, s_cursor, vmin, jp0_half, rowmask, cnt, edges, cap_edges,
                             cursor2, stream);
}
extern "C" int ssg_region_query_s_dev(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, const double* eps_dev, const uint32_t* s_pool,
                                      const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                                      uint16_t jp0_half, uint8_t* rowmask, int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor2,
                                      hipStream_t stream) {
  if (!eps_dev) { ssg_set_error("ssg_region_query_s_dev: null eps"); return SSG_ERR_INVALID; }
  return region_query_s_impl(M, v, N, row0, nrows, lambda_value, 0.0, eps_dev, s_pool, seg_off, seg_len, nseg, s_cursor, vmin, jp0_half, rowmask, cnt, edges, cap_edges,
                             cursor2, stream);
}

// workspace: parent[N] int32 | lab[N] int32 | rootflag[N] int32 | rootid[N+1] int64
extern "C" size_t ssg_dbscan_cc_workspace_bytes(int N) { return (size_t)N * 12 + ((size_t)N + 1) * 8 + 64; }

static int dbscan_cc_impl(const int32_t* cnt, const int32_t* edges, uint64_t nedges, const unsigned long long* ne_dev, int N, int min_samples, void* ws,
                          size_t ws_bytes, int64_t* labels, hipStream_t stream) {
  if (N <= 0 || ws_bytes < ssg_dbscan_cc_workspace_bytes(N)) { ssg_set_error("ssg_dbscan_cc: workspace too small"); return SSG_ERR_INVALID; }
  int* parent = (int*)ws; int* lab = parent + N; int32_t* rootflag = lab + N;
  int64_t* rootid = (int64_t*)(((uintptr_t)(rootflag + N) + 15) & ~(uintptr_t)15);
  const int nb = (N + 255) / 256;
  const int eb = nedges ? (int)((nedges + 255) / 256 < 8192 ? (nedges + 255) / 256 : 8192) : 1;
  hipLaunchKernelGGL(cc_init_kernel, dim3(nb), dim3(256), 0, stream, cnt, N, min_samples, parent, lab);
  if (nedges) {
    hipLaunchKernelGGL(cc_hook_kernel, dim3(eb), dim3(256), 0, stream, edges, (unsigned long long)nedges, ne_dev, cnt, min_samples, parent);
    hipLaunchKernelGGL(cc_union_kernel, dim3(eb), dim3(256), 0, stream, edges, (unsigned long long)nedges, ne_dev, cnt, min_samples, parent);
  }
  hipLaunchKernelGGL(cc_flatten_kernel, dim3(nb), dim3(256), 0, stream, cnt, N, min_samples, parent, rootflag);
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, stream, rootflag, N, rootid);
  if (nedges) hipLaunchKernelGGL(cc_border_kernel, dim3(eb > nb ? eb : nb), dim3(256), 0, stream, edges, (unsigned long long)nedges, ne_dev, cnt, min_samples, parent, rootid, lab, N);
  else hipLaunchKernelGGL(cc_label_core_kernel, dim3(nb), dim3(256), 0, stream, cnt, N, min_samples, parent, rootid, lab);
  // labels == NULL (round 6): the caller reads lab = (int32*)ws + N itself (0x7fffffff = noise) -- one launch less in a chain that reads the workspace back anyway
  if (labels) hipLaunchKernelGGL(cc_finalize_kernel, dim3(nb), dim3(256), 0, stream, lab, N, labels);
  SSG_LAUNCH_CHECK("dbscan_cc");
  return SSG_OK;
}

extern "C" int ssg_dbscan_cc(const int32_t* cnt, const int32_t* edges, uint64_t nedges, int N, int min_samples, void* ws, size_t ws_bytes,
                             int64_t* labels, hipStream_t stream) {
  return dbscan_cc_impl(cnt, edges, nedges, nullptr, N, min_samples, ws, ws_bytes, labels, stream);
}
// the same with the edge count left on the device: edges holds min(*nedges_dev, cap_edges) entries (the region query's cursor
// and capacity), so region query -> components -> labels runs without a host round trip in between
extern "C" int ssg_dbscan_cc_dev(const int32_t* cnt, const int32_t* edges, const uint64_t* nedges_dev, uint64_t cap_edges, int N, int min_samples,
                                 void* ws, size_t ws_bytes, int64_t* labels, hipStream_t stream) {
  if (!nedges_dev) { ssg_set_error("ssg_dbscan_cc_dev: null edge counter"); return SSG_ERR_INVALID; }
  return dbscan_cc_impl(cnt, edges, cap_edges, (const unsigned long long*)nedges_dev, N, min_samples, ws, ws_bytes, labels, stream);
}
