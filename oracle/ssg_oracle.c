/*
 * ssg_oracle.c -- CPU restatement of the SSG pseudo-label grouping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (self-similarity-grouping_amd/) never imports it and fails loudly without its
 * HIP extension.
 *
 * Every function restates one stage of the reference (paths relative to
 * /root/reference) with the reference's exact rounding points:
 *   reid/rerank.py:27-127      re_ranking (SSG fp16 k-reciprocal re-rank)
 *   selftraining.py:289-293    epsilon rule
 *   selftraining.py:295,306    sklearn.cluster.DBSCAN(metric='precomputed') 1.7.2
 *                              (third-party, not vendored; algorithm restated from
 *                              sklearn/cluster/_dbscan.py + _dbscan_inner.pyx)
 * numpy semantics restated here were probed against numpy 2.2.6 (see
 * tools/make_golden.py and tests/test_oracle_golden.py):
 *   - half arithmetic = float32 op then round-to-nearest-even to half
 *   - np.sum on half/float = pairwise summation (8 accumulators, blocks of 128)
 *   - np.mean(half, axis=0) = sequential float32 sum over rows, /n, -> half
 *   - np.argsort default kind = introsort (aquicksort), unstable
 *   - np.exp(half): this oracle uses the correctly rounded half result; numpy on an
 *     AVX512-FP16 host differs from that on 2 inputs in [0,8] (DESIGN.md "exp").
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: scipy's cdist accumulates s += d*d without FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ half <-> float/double */
static inline float h2f(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, b;
  if (e == 0) {
    if (m == 0) b = s;
    else { int sh = 0; while (!(m & 0x400)) { m <<= 1; sh++; } m &= 0x3ff; b = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
  } else if (e == 31) b = s | 0x7f800000u | (m << 13);
  else b = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &b, 4); return f;
}
/* direct double -> half, round-to-nearest-even (npy_double_to_half semantics) */
static inline uint16_t d2h(double d) {
  uint64_t b; memcpy(&b, &d, 8);
  uint16_t sign = (uint16_t)((b >> 48) & 0x8000);
  int e = (int)((b >> 52) & 0x7ff);
  uint64_t m = b & 0xfffffffffffffULL;
  if (e == 0x7ff) return (uint16_t)(sign | 0x7c00 | (m ? 0x200 : 0));
  if (e == 0) return sign;
  int he = e - 1023 + 15;
  if (he >= 31) return (uint16_t)(sign | 0x7c00);
  m |= 1ULL << 52;
  int shift = 42;
  if (he <= 0) { shift = 43 - he; if (shift > 63) return sign; }
  uint64_t q = m >> shift, rem = m & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
  if (rem > half || (rem == half && (q & 1))) q++;
  if (he <= 0) return (uint16_t)(sign | (uint16_t)q);
  uint32_t r = ((uint32_t)he << 10) + (uint32_t)(q - 0x400);
  if (r >= 0x7c00) r = 0x7c00;
  return (uint16_t)(sign | r);
}
static inline uint16_t f2h(float f) { return d2h((double)f); }
static inline int h_isnan(uint16_t h) { return (h & 0x7fff) > 0x7c00; }
/* numpy half_tag::less : a<b, NaNs sort last */
static inline int h_less(uint16_t a, uint16_t b) {
  if (h_isnan(b)) return !h_isnan(a);
  if (h_isnan(a)) return 0;
  return h2f(a) < h2f(b);
}

/* half binary ops = float op + RNE (numpy HALF_add/subtract/multiply/divide loops) */
static inline uint16_t h_add(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); }
static inline uint16_t h_sub(uint16_t a, uint16_t b) { return f2h(h2f(a) - h2f(b)); }
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }
static inline uint16_t h_div(uint16_t a, uint16_t b) { return f2h(h2f(a) / h2f(b)); }
#define H_ONE 0x3c00
#define H_TWO 0x4000

/* correctly rounded half exp(x) for every half x; table built once */
static uint16_t g_exp16[65536];
static int g_exp16_ready = 0;
static void build_exp16(void) {
  if (g_exp16_ready) return;
  for (uint32_t i = 0; i < 65536; i++) g_exp16[i] = d2h(exp((double)h2f((uint16_t)i)));
  g_exp16_ready = 1;
}
static inline uint16_t h_exp(uint16_t x) { return g_exp16[x]; }
static inline uint16_t h_neg(uint16_t x) { return (uint16_t)(x ^ 0x8000); }

void ora_half_exp_table(uint16_t* out) { build_exp16(); memcpy(out, g_exp16, sizeof(g_exp16)); }
void ora_f32_to_f16(const float* in, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = f2h(in[i]); }
void ora_f64_to_f16(const double* in, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = d2h(in[i]); }
void ora_f16_to_f32(const uint16_t* in, float* out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = h2f(in[i]); }

/* numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src @TYPE@_pairwise_sum),
 * float accumulators; used by np.sum on half and float arrays. */
static float pairwise_sum_f32(const float* a, int64_t n) {
  if (n < 8) { float r = 0.f; for (int64_t i = 0; i < n; i++) r += a[i]; return r; }
  if (n <= 128) {
    float r[8]; int64_t i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int64_t n2 = n / 2; n2 -= n2 % 8;
  return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
}
static double pairwise_sum_f64(const double* a, int64_t n) {
  if (n < 8) { double r = 0.; for (int64_t i = 0; i < n; i++) r += a[i]; return r; }
  if (n <= 128) {
    double r[8]; int64_t i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int64_t n2 = n / 2; n2 -= n2 % 8;
  return pairwise_sum_f64(a, n2) + pairwise_sum_f64(a + n2, n - n2);
}
float ora_pairwise_sum_f32(const float* a, int64_t n) { return pairwise_sum_f32(a, n); }
double ora_pairwise_sum_f64(const double* a, int64_t n) { return pairwise_sum_f64(a, n); }

/* scipy.spatial.distance.cdist(..., 'euclidean') inner loop (scipy/spatial/src/
 * distance_impl.h sqeuclidean_distance_double): sequential s += d*d in double. */
static inline double sqeuclid_seq(const double* u, const double* v, int d) {
  double s = 0.0;
  for (int i = 0; i < d; i++) { const double t = u[i] - v[i]; s += t * t; }
  return s;
}

/* ------------------------------------------------------------------ rerank.py:35-40
 * v_i = min_s f16(1 - exp(-f16(cdist(tgt,src)^2)));  v /= max(v)   (all half)
 * tgt [N,d] f32, src [Ns,d] f32 -> v_raw[N] (before /max), v[N] (after).  Returns
 * the half max (0 => the reference divides 0/0 -> NaN, rerank.py:40). */
uint16_t ora_source_vec(const float* tgt, const float* src, int N, int Ns, int d, uint16_t* v_raw, uint16_t* v) {
  build_exp16();
  double* S = (double*)malloc((size_t)Ns * d * sizeof(double));
  for (int64_t i = 0; i < (int64_t)Ns * d; i++) S[i] = (double)src[i];
#pragma omp parallel
  {
    double* t = (double*)malloc((size_t)d * sizeof(double));
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < d; k++) t[k] = (double)tgt[(int64_t)i * d + k];
      uint16_t best = 0; int have = 0;
      for (int s = 0; s < Ns; s++) {
        double dist = sqrt(sqeuclid_seq(t, S + (int64_t)s * d, d));
        uint16_t h = d2h(dist * dist);                 /* np.power(cdist,2).astype(f16)   :36-37 */
        uint16_t o = h_sub(H_ONE, h_exp(h_neg(h)));    /* 1-np.exp(-x)                     :38    */
        if (!have) { best = o; have = 1; }                /* np.min propagates NaN :39 */
        else if (h_isnan(best)) { }
        else if (h_isnan(o) || h2f(o) < h2f(best)) best = o;
      }
      v_raw[i] = best;
    }
    free(t);
  }
  free(S);
  uint16_t mx = v_raw[0];
  for (int i = 1; i < N; i++) if (h_isnan(v_raw[i]) || (!h_isnan(mx) && h2f(v_raw[i]) > h2f(mx))) mx = v_raw[i];
  for (int i = 0; i < N; i++) v[i] = h_div(v_raw[i], mx);   /* :40 */
  return mx;
}

/* ------------------------------------------------------------------ rerank.py:33,61-62
 * feat = f16(x); D = f16( f16(cdist_f64(feat,feat))^2 )  -> D [N,N] half
 * memory_save (rerank.py:49-59, MemorySave=True): D = f16( cdist_f64(feat,feat)^2 ), squared in float64, one rounding
 * (the Minibatch row chunks of that branch do not change any value). */
void ora_euclid2(const float* tgt, int N, int d, int memory_save, uint16_t* D) {
  double* F = (double*)malloc((size_t)N * d * sizeof(double));
  for (int64_t i = 0; i < (int64_t)N * d; i++) F[i] = (double)h2f(f2h(tgt[i]));
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      double sq = sqrt(sqeuclid_seq(F + (int64_t)i * d, F + (int64_t)j * d, d));
      uint16_t h = d2h(sq);
      D[(int64_t)i * N + j] = memory_save ? d2h(sq * sq) : h_mul(h, h);  /* np.power(half,2) = half(float(h)*float(h)) */
    }
  free(F);
}
void ora_euclid(const float* tgt, int N, int d, uint16_t* D) { ora_euclid2(tgt, N, d, 0, D); }

/* numpy npysort/quicksort.cpp aquicksort_<half> + heapsort.cpp aheapsort_ (published
 * algorithm, restated): introsort on an index array, unstable. */
static void aheapsort_half(const uint16_t* v, int64_t* tosort, int64_t n) {
  int64_t *a = tosort - 1, i, j, l, tmp;
  for (l = n >> 1; l > 0; --l) {
    tmp = a[l];
    for (i = l, j = l << 1; j <= n;) {
      if (j < n && h_less(v[a[j]], v[a[j + 1]])) j += 1;
      if (h_less(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; } else break;
    }
    a[i] = tmp;
  }
  for (; n > 1;) {
    tmp = a[n]; a[n] = a[1]; n -= 1;
    for (i = 1, j = 2; j <= n;) {
      if (j < n && h_less(v[a[j]], v[a[j + 1]])) j++;
      if (h_less(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; } else break;
    }
    a[i] = tmp;
  }
}
static int msb64(uint64_t n) { int k = 0; while (n >>= 1) k++; return k; }
#define SWAPI(a, b) do { int64_t t_ = (a); (a) = (b); (b) = t_; } while (0)
/* small_thr: ranges with pr - pl > small_thr are partitioned.  numpy's source says SMALL_QUICKSORT = 16, the numpy 2.2.6 wheel
 * in this image behaves like 15 (17 elements are still partitioned; probed, and pinned by tests/test_oracle_golden.py on the
 * installed numpy): the tie order of reid/rerank.py:70 therefore depends on the numpy build, see DESIGN.md "Ties". */
static int g_small_thr = 15;
void ora_set_small_threshold(int t) { g_small_thr = t; }
static void aquicksort_half(const uint16_t* v, int64_t* tosort, int64_t num) {
  int64_t *pl = tosort, *pr = tosort + num - 1, *stack[128], **sptr = stack, *pm, *pi, *pj, *pk, vi;
  int depth[128], *psdepth = depth, cdepth = msb64((uint64_t)num) * 2;
  uint16_t vp;
  for (;;) {
    if (cdepth < 0) { aheapsort_half(v, pl, pr - pl + 1); goto stack_pop; }
    while ((pr - pl) > g_small_thr) {   /* numpy 2.2.6 behaviour (15): 17 elements are still partitioned (probed) */
      pm = pl + ((pr - pl) >> 1);
      if (h_less(v[*pm], v[*pl])) SWAPI(*pm, *pl);
      if (h_less(v[*pr], v[*pm])) SWAPI(*pr, *pm);
      if (h_less(v[*pm], v[*pl])) SWAPI(*pm, *pl);
      vp = v[*pm]; pi = pl; pj = pr - 1; SWAPI(*pm, *pj);
      for (;;) {
        do { ++pi; } while (h_less(v[*pi], vp));
        do { --pj; } while (h_less(vp, v[*pj]));
        if (pi >= pj) break;
        SWAPI(*pi, *pj);
      }
      pk = pr - 1; SWAPI(*pi, *pk);
      if (pi - pl < pr - pi) { *sptr++ = pi + 1; *sptr++ = pr; pr = pi - 1; }
      else { *sptr++ = pl; *sptr++ = pi - 1; pl = pi + 1; }
      *psdepth++ = --cdepth;
    }
    for (pi = pl + 1; pi <= pr; ++pi) {
      vi = *pi; vp = v[vi]; pj = pi; pk = pi - 1;
      while (pj > pl && h_less(vp, v[*pk])) { *pj-- = *pk--; }
      *pj = vi;
    }
  stack_pop:
    if (sptr == stack) break;
    pr = *(--sptr); pl = *(--sptr); cdepth = *(--psdepth);
  }
}
void ora_argsort_half(const uint16_t* v, int64_t n, int64_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = i;
  aquicksort_half(v, out, n);
}

/* ------------------------------------------------------------------ rerank.py:68-70
 * Dn = transpose(D / max(D, axis=0));  initial_rank = argsort(Dn)[:, :K]
 * mode 0: canonical (value, index) order (== numpy kind='stable');
 * mode 1: numpy default introsort (bit-parity with the unpatched reference incl. ties).
 * Outputs: Dn [N,N] half, rank [N,K] int32, colmax [N] half. */
void ora_normalize_rank(const uint16_t* D, int N, int K, int mode, uint16_t* Dn, int32_t* rank, uint16_t* colmax) {
  if (K > N) K = N;
  for (int j = 0; j < N; j++) {
    uint16_t m = D[j];
    for (int i = 1; i < N; i++) { uint16_t x = D[(int64_t)i * N + j]; if (h_isnan(x) || (!h_isnan(m) && h2f(x) > h2f(m))) m = x; }
    colmax[j] = m;
  }
#pragma omp parallel
  {
    int64_t* idx = (int64_t*)malloc((size_t)N * sizeof(int64_t));
#pragma omp for schedule(dynamic, 16)
    for (int a = 0; a < N; a++) {
      uint16_t* row = Dn + (int64_t)a * N;
      for (int b = 0; b < N; b++) row[b] = h_div(D[(int64_t)b * N + a], colmax[a]);
      if (mode == 1) {
        ora_argsort_half(row, N, idx);
        for (int r = 0; r < K; r++) rank[(int64_t)a * K + r] = (int32_t)idx[r];
      } else {
        /* partial selection of the K smallest by (value, index); NaNs last */
        int cnt = 0; int32_t* out = rank + (int64_t)a * K;
        for (int b = 0; b < N; b++) {
          if (cnt == K && !h_less(row[b], row[out[K - 1]])) continue;
          int p = cnt < K ? cnt : K - 1;
          while (p > 0 && h_less(row[b], row[out[p - 1]])) { out[p] = out[p - 1]; p--; }
          out[p] = b; if (cnt < K) cnt++;
        }
      }
    }
    free(idx);
  }
}

/* ------------------------------------------------------------------ rerank.py:74-92
 * k-reciprocal sets, 1/2-k expansion, Gaussian weights -> dense V [N,N] half (zeros elsewhere) */
static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }
void ora_krecip(const uint16_t* Dn, const int32_t* rank, int N, int K, int k1, uint16_t* V) {
  build_exp16();
  int K1 = k1 + 1; if (K1 > N) K1 = N; if (K1 > K) K1 = K;
  int kh = (int)rint((double)k1 / 2.0) + 1; if (kh > N) kh = N; if (kh > K) kh = K;   /* int(np.around(k1/2))+1 */
  memset(V, 0, (size_t)N * N * sizeof(uint16_t));
#pragma omp parallel
  {
    int cap = K1 + K1 * kh + 8;
    int32_t* rec = (int32_t*)malloc(sizeof(int32_t) * (K1 + 1));
    int32_t* crec = (int32_t*)malloc(sizeof(int32_t) * (kh + 1));
    int32_t* expn = (int32_t*)malloc(sizeof(int32_t) * cap);
    float* w = (float*)malloc(sizeof(float) * cap);
    uint16_t* wh = (uint16_t*)malloc(sizeof(uint16_t) * cap);
#pragma omp for schedule(dynamic, 32)
    for (int i = 0; i < N; i++) {
      const int32_t* fwd = rank + (int64_t)i * K;
      int nrec = 0;
      for (int a = 0; a < K1; a++) {            /* :76-79 */
        const int32_t* bw = rank + (int64_t)fwd[a] * K; int hit = 0;
        for (int b = 0; b < K1; b++) if (bw[b] == i) { hit = 1; break; }
        if (hit) rec[nrec++] = fwd[a];
      }
      int ne = 0;
      for (int a = 0; a < nrec; a++) expn[ne++] = rec[a];
      for (int a = 0; a < nrec; a++) {          /* :81-88 */
        int32_t cand = rec[a]; const int32_t* cf = rank + (int64_t)cand * K; int nc = 0;
        for (int b = 0; b < kh; b++) {
          const int32_t* cb = rank + (int64_t)cf[b] * K; int hit = 0;
          for (int c = 0; c < kh; c++) if (cb[c] == cand) { hit = 1; break; }
          if (hit) crec[nc++] = cf[b];
        }
        int inter = 0;
        for (int b = 0; b < nc; b++) for (int c = 0; c < nrec; c++) if (crec[b] == rec[c]) { inter++; break; }
        if ((double)inter > (2.0 / 3.0) * (double)nc) for (int b = 0; b < nc; b++) expn[ne++] = crec[b];
      }
      qsort(expn, ne, sizeof(int32_t), cmp_i32);  /* np.unique :90 */
      int nu = 0;
      for (int a = 0; a < ne; a++) if (nu == 0 || expn[a] != expn[nu - 1]) expn[nu++] = expn[a];
      const uint16_t* drow = Dn + (int64_t)i * N;
      for (int a = 0; a < nu; a++) { wh[a] = h_exp(h_neg(drow[expn[a]])); w[a] = h2f(wh[a]); }  /* :91 */
      uint16_t sum = f2h(pairwise_sum_f32(w, nu));                                              /* np.sum -> half */
      for (int a = 0; a < nu; a++) V[(int64_t)i * N + expn[a]] = h_div(wh[a], sum);             /* :92 */
    }
    free(rec); free(crec); free(expn); free(w); free(wh);
  }
}

/* ------------------------------------------------------------------ rerank.py:94-99
 * V_qe[i,:] = half( float32 sum_{r<k2} V[rank[i,r],:] / k2 ) */
void ora_query_expansion(const uint16_t* V, const int32_t* rank, int N, int K, int k2, uint16_t* Vqe) {
  int kk = k2; if (kk > N) kk = N; if (kk > K) kk = K;
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < N; i++) {
    const int32_t* r = rank + (int64_t)i * K;
    for (int j = 0; j < N; j++) {
      float s = h2f(V[(int64_t)r[0] * N + j]);
      for (int q = 1; q < kk; q++) s += h2f(V[(int64_t)r[q] * N + j]);
      Vqe[(int64_t)i * N + j] = f2h(s / (float)kk);
    }
  }
}

/* ------------------------------------------------------------------ rerank.py:101-119
 * inverted index + Jaccard (sequential half adds in ascending column order) + clamp */
void ora_jaccard(const uint16_t* V, int N, uint16_t* J) {
  int64_t* colptr = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
  for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) if (V[(int64_t)r * N + c] & 0x7fff) colptr[c + 1]++;
  for (int c = 0; c < N; c++) colptr[c + 1] += colptr[c];
  int32_t* rows = (int32_t*)malloc((size_t)(colptr[N] ? colptr[N] : 1) * sizeof(int32_t));
  int64_t* fill = (int64_t*)malloc((size_t)N * sizeof(int64_t));
  memcpy(fill, colptr, (size_t)N * sizeof(int64_t));
  for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) if (V[(int64_t)r * N + c] & 0x7fff) rows[fill[c]++] = r;  /* ascending r :103 */
#pragma omp parallel
  {
    uint16_t* t = (uint16_t*)malloc((size_t)N * sizeof(uint16_t));
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < N; i++) {
      memset(t, 0, (size_t)N * sizeof(uint16_t));
      for (int c = 0; c < N; c++) {                /* indNonZero ascending :110 */
        uint16_t vic = V[(int64_t)i * N + c];
        if (!(vic & 0x7fff)) continue;
        for (int64_t p = colptr[c]; p < colptr[c + 1]; p++) {
          int r = rows[p]; uint16_t vrc = V[(int64_t)r * N + c];
          uint16_t mn = h2f(vrc) < h2f(vic) ? vrc : vic;      /* np.minimum :114 */
          t[r] = h_add(t[r], mn);
        }
      }
      for (int k = 0; k < N; k++) {                 /* 1-temp_min/(2-temp_min) :115 ; clamp :117-118 */
        uint16_t j = h_sub(H_ONE, h_div(t[k], h_sub(H_TWO, t[k])));
        if (h2f(j) < 0.f) j = 0;
        J[(int64_t)i * N + k] = j;
      }
    }
    free(t);
  }
  free(colptr); free(rows); free(fill);
}

/* ------------------------------------------------------------------ rerank.py:122
 * final = half(J * half(1-lambda)) (as f64) + f64(half(v_i+v_k)) * lambda
 * Jp (optional, may be NULL) receives the compact half J' */
void ora_final(const uint16_t* J, const uint16_t* v, int N, double lambda_value, double* final_dist, uint16_t* Jp) {
  uint16_t om = d2h(1.0 - lambda_value);   /* python float (weak scalar) -> half under NEP 50 */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++)
    for (int k = 0; k < N; k++) {
      uint16_t jp = h_mul(J[(int64_t)i * N + k], om);
      if (Jp) Jp[(int64_t)i * N + k] = jp;
      final_dist[(int64_t)i * N + k] = (double)h2f(jp) + (double)h2f(h_add(v[k], v[i])) * lambda_value;
    }
}

/* ------------------------------------------------------------------ selftraining.py:289-293
 * eps = mean of the round(rho*count) smallest non-zero entries of the strict upper triangle.
 * f64 matrix: np.mean = pairwise f64 sum / n.  Returns NaN when top_num == 0 (numpy mean of
 * an empty slice). count_out / top_out report the two integers. */
static int cmp_f64(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }
double ora_eps_f64(const double* M, int N, double rho, int64_t* count_out, int64_t* top_out) {
  int64_t cap = (int64_t)N * (N - 1) / 2, cnt = 0;
  double* t = (double*)malloc((size_t)(cap ? cap : 1) * sizeof(double));
  for (int i = 0; i < N; i++) for (int k = i + 1; k < N; k++) { double x = M[(int64_t)i * N + k]; if (x != 0.0) t[cnt++] = x; }
  qsort(t, cnt, sizeof(double), cmp_f64);
  int64_t top = (int64_t)rint(rho * (double)cnt);   /* np.round: half to even */
  if (count_out) *count_out = cnt;
  if (top_out) *top_out = top;
  double eps = top > 0 ? pairwise_sum_f64(t, top) / (double)top : NAN;
  free(t);
  return eps;
}
/* half matrix (the no-rerank euclidean_dist): np.mean(half) = float32 pairwise sum, /n in
 * float64 (float32 scalar / intp scalar promotes), then np.float16(...) directly.  Returns the half bits. */
static int cmp_h(const void* a, const void* b) { float x = h2f(*(const uint16_t*)a), y = h2f(*(const uint16_t*)b); return (x > y) - (x < y); }
uint16_t ora_eps_f16(const uint16_t* M, int N, double rho, int64_t* count_out, int64_t* top_out) {
  int64_t cap = (int64_t)N * (N - 1) / 2, cnt = 0;
  uint16_t* t = (uint16_t*)malloc((size_t)(cap ? cap : 1) * sizeof(uint16_t));
  for (int i = 0; i < N; i++) for (int k = i + 1; k < N; k++) { uint16_t x = M[(int64_t)i * N + k]; if (x & 0x7fff) t[cnt++] = x; }
  qsort(t, cnt, sizeof(uint16_t), cmp_h);
  int64_t top = (int64_t)rint(rho * (double)cnt);
  if (count_out) *count_out = cnt;
  if (top_out) *top_out = top;
  uint16_t eps = 0x7e00;
  if (top > 0) {
    float* f = (float*)malloc((size_t)top * sizeof(float));
    for (int64_t i = 0; i < top; i++) f[i] = h2f(t[i]);
    float s = pairwise_sum_f32(f, top);
    eps = d2h((double)s / (double)top);   /* np.float32 scalar / np.intp scalar -> float64, then np.float16(...) */
    free(f);
  }
  free(t);
  return eps;
}

/* ------------------------------------------------------------------ sklearn 1.7.2 DBSCAN,
 * metric='precomputed' (cluster/_dbscan.py fit + neighbors/_base.py radius_neighbors
 * `d <= radius` + cluster/_dbscan_inner.pyx).  M [N,N] f64, row-wise neighbourhoods,
 * depth-first expansion in index order with a LIFO stack exactly as dbscan_inner. */
void ora_dbscan(const double* M, int N, double eps, int min_samples, int64_t* labels) {
  int64_t* nptr = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
  for (int i = 0; i < N; i++) { int64_t c = 0; for (int k = 0; k < N; k++) c += (M[(int64_t)i * N + k] <= eps); nptr[i + 1] = nptr[i] + c; }
  int32_t* nb = (int32_t*)malloc((size_t)(nptr[N] ? nptr[N] : 1) * sizeof(int32_t));
  for (int i = 0; i < N; i++) { int64_t p = nptr[i]; for (int k = 0; k < N; k++) if (M[(int64_t)i * N + k] <= eps) nb[p++] = k; }
  uint8_t* core = (uint8_t*)malloc((size_t)N);
  for (int i = 0; i < N; i++) { core[i] = (nptr[i + 1] - nptr[i]) >= min_samples; labels[i] = -1; }
  int32_t* stack = (int32_t*)malloc((size_t)(nptr[N] + N + 1) * sizeof(int32_t));
  int64_t label_num = 0;
  for (int i0 = 0; i0 < N; i0++) {
    if (labels[i0] != -1 || !core[i0]) continue;
    int64_t sp = 0; int i = i0;
    for (;;) {
      if (labels[i] == -1) {
        labels[i] = label_num;
        if (core[i]) for (int64_t p = nptr[i]; p < nptr[i + 1]; p++) { int v = nb[p]; if (labels[v] == -1) stack[sp++] = v; }
      }
      if (sp == 0) break;
      i = stack[--sp];
    }
    label_num++;
  }
  free(nptr); free(nb); free(core); free(stack);
}

/* ------------------------------------------------------------------ one-call pipeline,
 * used as bench.py's cpu_baseline ("port") and by tests that want every stage boundary.
 * Any output pointer may be NULL.  Returns 0, or 1 if max(source_dist_vec)==0 (NaN path). */
int ora_re_ranking(const float* src, const float* tgt, int Ns, int N, int d, int k1, int k2, double lambda_value,
                   int rank_mode, int memory_save, uint16_t* euclid /*[N,N]*/, uint16_t* v_out /*[N]*/,
                   int32_t* rank_out /*[N,min(N,max(k1+1,k2))]*/,
                   uint16_t* V_out, uint16_t* Vqe_out, uint16_t* J_out, uint16_t* Jp_out, double* final_out) {
  int K = k1 + 1; if (k2 > K) K = k2; if (K > N) K = N;   /* columns of initial_rank that are ever read (:76,:83,:97) */
  size_t nn = (size_t)N * N;
  uint16_t* v_raw = (uint16_t*)malloc((size_t)N * 2); uint16_t* v = (uint16_t*)malloc((size_t)N * 2);
  uint16_t mx = ora_source_vec(tgt, src, N, Ns, d, v_raw, v);
  if (v_out) memcpy(v_out, v, (size_t)N * 2);
  uint16_t* D = euclid ? euclid : (uint16_t*)malloc(nn * 2);
  ora_euclid2(tgt, N, d, memory_save, D);
  uint16_t* Dn = (uint16_t*)malloc(nn * 2); uint16_t* colmax = (uint16_t*)malloc((size_t)N * 2);
  int32_t* rank = rank_out ? rank_out : (int32_t*)malloc((size_t)N * K * 4);
  ora_normalize_rank(D, N, K, rank_mode, Dn, rank, colmax);
  uint16_t* V = V_out ? V_out : (uint16_t*)malloc(nn * 2);
  ora_krecip(Dn, rank, N, K, k1, V);
  uint16_t* Vq = V;
  if (k2 != 1) { Vq = Vqe_out ? Vqe_out : (uint16_t*)malloc(nn * 2); ora_query_expansion(V, rank, N, K, k2, Vq); }
  uint16_t* J = J_out ? J_out : (uint16_t*)malloc(nn * 2);
  ora_jaccard(Vq, N, J);
  if (final_out) ora_final(J, v, N, lambda_value, final_out, Jp_out);
  if (!J_out) free(J);
  if (Vq != V && !Vqe_out) free(Vq);
  if (!V_out) free(V);
  if (!rank_out) free(rank);
  free(Dn); free(colmax); if (!euclid) free(D); free(v_raw); free(v);
  return (mx & 0x7fff) == 0;
}

/* ------------------------------------------------------------------ reid/rerank.py:171-234 ==
 * reid/rerank_initial.py:40-99  re_ranking_init (float32 cosine variant, caller reid/eug.py:223-226).
 * dots [N,N] float32 = the stacked [[q_q, q_g],[q_g^T, g_g]] dot products (np.dot in the reference; the
 * caller of this oracle provides them so that BLAS summation order is not part of the restatement).
 * out [nq, N-nq] float32.  np.exp(float32) is libm-class accurate here (numpy's SIMD exp differs in the
 * last ulp): parity for this variant is tolerance based. */
static int cmp_pair_f(const void* a, const void* b) {
  const float x = ((const float*)a)[0], y = ((const float*)b)[0];
  if (x < y) return -1; if (x > y) return 1;
  const float i = ((const float*)a)[1], j = ((const float*)b)[1];
  return (i > j) - (i < j);
}
void ora_re_ranking_init(const float* dots, int N, int nq, int k1, int k2, float lambda_value, float* out) {
  const size_t nn = (size_t)N * N;
  float* od = (float*)malloc(nn * 4);            /* original_dist = 2 - 2*dots, then transpose(od / max(od, axis=0)) */
  float* Dn = (float*)malloc(nn * 4);
  for (size_t x = 0; x < nn; x++) od[x] = 2.f - 2.f * dots[x];
  float* colmax = (float*)malloc((size_t)N * 4);
  for (int j = 0; j < N; j++) { float m = od[j]; for (int i = 1; i < N; i++) if (od[(size_t)i * N + j] > m) m = od[(size_t)i * N + j]; colmax[j] = m; }
  for (int a = 0; a < N; a++) for (int b = 0; b < N; b++) Dn[(size_t)a * N + b] = 1.f * od[(size_t)b * N + a] / colmax[a];
  int K = k1 + 1; if (K > N) K = N;
  int kh = (int)rint((double)k1 / 2.0) + 1; if (kh > K) kh = K;
  int32_t* rank = (int32_t*)malloc((size_t)N * K * 4);
#pragma omp parallel
  {
    float* pr = (float*)malloc((size_t)N * 8);
#pragma omp for schedule(dynamic, 16)
    for (int a = 0; a < N; a++) {              /* argpartition(range(1,k1+1)): k1+1 smallest ascending */
      for (int b = 0; b < N; b++) { pr[2 * b] = Dn[(size_t)a * N + b]; pr[2 * b + 1] = (float)b; }
      qsort(pr, N, 8, cmp_pair_f);
      for (int r = 0; r < K; r++) rank[(size_t)a * K + r] = (int32_t)pr[2 * r + 1];
    }
    free(pr);
  }
  float* V = (float*)calloc(nn, 4);
#pragma omp parallel
  {
    int cap = K + K * kh + 8;
    int32_t* rec = (int32_t*)malloc(4 * (K + 1)); int32_t* crec = (int32_t*)malloc(4 * (kh + 1)); int32_t* expn = (int32_t*)malloc(4 * cap);
    float* w = (float*)malloc(4 * cap);
#pragma omp for schedule(dynamic, 32)
    for (int i = 0; i < N; i++) {
      const int32_t* fwd = rank + (size_t)i * K; int nrec = 0;
      for (int a = 0; a < K; a++) { const int32_t* bw = rank + (size_t)fwd[a] * K; int hit = 0; for (int b = 0; b < K; b++) if (bw[b] == i) { hit = 1; break; } if (hit) rec[nrec++] = fwd[a]; }
      int ne = 0; for (int a = 0; a < nrec; a++) expn[ne++] = rec[a];
      for (int a = 0; a < nrec; a++) {
        int32_t cand = rec[a]; const int32_t* cf = rank + (size_t)cand * K; int nc = 0;
        for (int b = 0; b < kh; b++) { const int32_t* cb = rank + (size_t)cf[b] * K; int hit = 0; for (int c = 0; c < kh; c++) if (cb[c] == cand) { hit = 1; break; } if (hit) crec[nc++] = cf[b]; }
        int inter = 0; for (int b = 0; b < nc; b++) for (int c = 0; c < nrec; c++) if (crec[b] == rec[c]) { inter++; break; }
        if ((double)inter > (2.0 / 3.0) * (double)nc) for (int b = 0; b < nc; b++) expn[ne++] = crec[b];
      }
      qsort(expn, ne, 4, cmp_i32);
      int nu = 0; for (int a = 0; a < ne; a++) if (nu == 0 || expn[a] != expn[nu - 1]) expn[nu++] = expn[a];
      for (int a = 0; a < nu; a++) w[a] = expf(-Dn[(size_t)i * N + expn[a]]);
      const float sum = pairwise_sum_f32(w, nu);
      for (int a = 0; a < nu; a++) V[(size_t)i * N + expn[a]] = 1.f * w[a] / sum;
    }
    free(rec); free(crec); free(expn); free(w);
  }
  float* Vq = V;
  if (k2 != 1) {
    int kk = k2; if (kk > K) kk = K;
    Vq = (float*)malloc(nn * 4);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; i++) {
      const int32_t* r = rank + (size_t)i * K;
      for (int j = 0; j < N; j++) { float s = V[(size_t)r[0] * N + j]; for (int q = 1; q < kk; q++) s += V[(size_t)r[q] * N + j]; Vq[(size_t)i * N + j] = s / (float)kk; }
    }
  }
  int64_t* colptr = (int64_t*)calloc((size_t)N + 1, 8);
  for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) if (Vq[(size_t)r * N + c] != 0.f) colptr[c + 1]++;
  for (int c = 0; c < N; c++) colptr[c + 1] += colptr[c];
  int32_t* rows = (int32_t*)malloc((size_t)(colptr[N] ? colptr[N] : 1) * 4);
  int64_t* fill = (int64_t*)malloc((size_t)N * 8); memcpy(fill, colptr, (size_t)N * 8);
  for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) if (Vq[(size_t)r * N + c] != 0.f) rows[fill[c]++] = r;
  const int ng = N - nq;
#pragma omp parallel
  {
    float* t = (float*)malloc((size_t)N * 4);
#pragma omp for schedule(dynamic, 8)
    for (int i = 0; i < nq; i++) {
      memset(t, 0, (size_t)N * 4);
      for (int c = 0; c < N; c++) {
        const float vic = Vq[(size_t)i * N + c]; if (vic == 0.f) continue;
        for (int64_t p = colptr[c]; p < colptr[c + 1]; p++) { const int r = rows[p]; const float vr = Vq[(size_t)r * N + c]; t[r] = t[r] + (vr < vic ? vr : vic); }
      }
      for (int g = 0; g < ng; g++) {
        const float j = 1.f - t[nq + g] / (2.f - t[nq + g]);
        out[(size_t)i * ng + g] = j * (1.f - lambda_value) + Dn[(size_t)i * N + nq + g] * lambda_value;
      }
    }
    free(t);
  }
  if (Vq != V) free(Vq);
  free(V); free(rank); free(colmax); free(Dn); free(od); free(colptr); free(rows); free(fill);
}

int ora_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ora_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
