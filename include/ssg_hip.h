/*
 * ssg_hip.h -- C ABI of libssg_hip.so: the MI355X (gfx950) kernels behind the SSG
 * pseudo-label grouping hot path (extract -> N x N re-rank -> eps -> DBSCAN).
 *
 * The reference (SHI-Labs/Self-Similarity-Grouping) has no FFI layer: its boundary for this
 * path is the Python call surface of selftraining.py / reid/.  This header is what a ctypes
 * stub in that Python binds (INTEGRATION.md shows the stub); every entry point names the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller (PyTorch
 *     or any HIP allocator) owns all buffers including workspaces;
 *   - calls are asynchronous on `stream` (a hipStream_t, 0 = default stream), re-entrant,
 *     and keep no global state;
 *   - return value: SSG_OK (0) or a negative SSG_ERR_* code; ssg_last_error() gives the
 *     message (thread-local).  No C++ exception crosses the boundary;
 *   - "half" = IEEE binary16 stored as uint16_t bit patterns; arithmetic on it follows
 *     numpy's half loops (float32 op + round-to-nearest-even), see DESIGN.md;
 *   - row-block arguments (row0, nrows) address rows [row0, row0+nrows) of an N-row
 *     problem: one GPU of a row-sharded job passes its own block, a single GPU passes
 *     (0, N).
 */
#ifndef SSG_HIP_H
#define SSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ssg_stream_t; /* == hipStream_t */

#define SSG_OK 0
#define SSG_ERR_INVALID (-1)  /* bad argument / shape */
#define SSG_ERR_HIP (-2)      /* HIP runtime error */
#define SSG_ERR_OVERFLOW (-3) /* caller-provided capacity exceeded */
#define SSG_ERR_NAN (-4)

const char* ssg_last_error(void);
int ssg_version(void);
/* half(x) with one rounding; used for the weak python scalar (1-lambda) of rerank.py:122 */
uint16_t ssg_double_to_half_bits(double d);

/* ---- K3/K4 pairwise distance (replaces scipy cdist in reid/rerank.py:36-37,61-62) ------ */
/* Value ranges of the two feature sets in one launch: out4 (device, 4 floats) = [max|a|, max|b|, max row norm of a, max row norm of b];
 * the norms are float32 upper bounds (inflated by 1e-5).  a [rows_a,d], b [rows_b,d] or NULL.  The host picks the digit count of
 * ssg_gram_i8_encode and the tolerance / operand scales of ssg_source_rowmin_filtered* from them (one read-back). */
int ssg_range_stats_f32(const float* a, int rows_a, const float* b, int rows_b, int d, float* out4, ssg_stream_t stream);
/* norms[i] = sum_k x[i,k]^2 in float64, accumulated exactly like the Gram kernel.
 * round_to_half != 0 first rounds x to half (rerank.py:33 feat = astype(float16)). */
int ssg_row_norms_f64(const float* x, int n, int d, int round_to_half, double* norms, ssg_stream_t stream);
/* D[il, j] = half(half(sqrt(|f16(x_i) - f16(x_j)|^2))^2) for rows i = row0+il (rerank.py:33,61-62)
 * rowmax[il] = max_j D[il, j] as half bits in a uint32 (rerank.py:68 max(original_dist, axis=0)).
 * x [N,d] f32 row-major, d % 4 == 0; norms from ssg_row_norms_f64(x, N, d, 1).
 * memory_save != 0: the reference's MemorySave=True branch (rerank.py:49-59) D = half(sqrt(.)^2), squared in float64 and rounded
 * once (Minibatch only chunks the rows there and does not change a value). */
int ssg_sqdist_self_f16(const float* x, const double* norms, int N, int d, int row0, int nrows, int memory_save, uint16_t* D, uint32_t* rowmax,
                        ssg_stream_t stream);
/* The same matrix as an EXACT integer Gram on the int8 matrix cores (csrc/gram_i8.hip): for half-rounded features in
 * [-1, 1] (L2-normalised embeddings) feat*2^24 is an integer and scipy's float64 squared distance is exact, so
 * d2*2^48 = |X_i|^2 + |X_j|^2 - 2<X_i,X_j> in int64, with the dot product on v_mfma_i32_32x32x32_i8 over ndigits balanced
 * radix-256 digits (3: |feat| <= 0.498, 9 digit products; 4: |feat| <= 1, 16), gives the identical value.
 * ssg_gram_i8_encode writes the digits (ssg_gram_i8_encoded_bytes bytes: whole 64-row panels, layout [row / 64][k block][row % 64]
 * [digit][32] since round 4 -- opaque to the caller, consumed by ssg_sqdist_self_i8 only) and the exact int64 norms and sets *flag when a
 * feature does not fit (caller zeroes *flag first and falls back to ssg_sqdist_self_f16 when it is set;
 * ssg_sqdist_self_i8 itself does nothing in that case).  d <= 16384. */
size_t ssg_gram_i8_encoded_bytes(int n, int d, int ndigits);
int ssg_gram_i8_encode(const float* x, int n, int d, int ndigits, void* E, int64_t* norms, int32_t* flag, ssg_stream_t stream);
int ssg_sqdist_self_i8(const void* E, const int64_t* norms, int N, int d, int ndigits, int row0, int nrows, int memory_save, uint16_t* D,
                       uint32_t* rowmax,
                       const int32_t* flag, ssg_stream_t stream);
/* rowmin[i] = min_s half(cdist(tgt_i, src_s)^2) as half bits in uint32 (rerank.py:36-37,39) */
int ssg_source_rowmin_f16(const float* tgt, const double* ntgt, const float* src, const double* nsrc, int nrows, int Ns, int d,
                          uint32_t* rowmin, ssg_stream_t stream);
/* Same result by filter-and-refine: a float32 MFMA pass bounds every target-source distance per row and
 * 8-source granule, float64 re-evaluates only the granules within `tol` of the row's bound (tol >= float32
 * error of the bound).  src has Ns_pad rows (rows >= Ns padding), d % 32 == 0, Ns_pad % 128 == 0;
 * ws = nrows + Ns_pad + nrows*Ns_pad/8 floats (row norms + the per-row bounds of every 8-source granule).
 * scale_t, scale_s > 0: the bound pass runs on the fp16 matrix cores over split-half (h8l8) copies of
 * tgt*scale_t and src*scale_s (powers of two with max|x|*scale < 65504; ws grows by (nrows + Ns_pad)*d floats)
 * and tol must cover that pass's error; 0 = float32 MFMA bound pass. */
int ssg_source_rowmin_filtered(const float* tgt, const float* src, int nrows, int Ns, int Ns_pad, int d, float tol, float scale_t,
                               float scale_s, float* ws, uint32_t* rowmin, ssg_stream_t stream);
/* The same with the bound pass as a plain fp16 GEMM (csrc/source_bound.hip: half copies of tgt*scale_t and src*scale_s, one
 * v_mfma_f32_32x32x16_f16 product per term, 2 bytes per operand element instead of 4); tol must additionally cover 2^-10 |x||y| per
 * dot product -- a few more granules are re-evaluated in float64, the result is the same.  Needs scale_t, scale_s > 0 (runs the
 * three-product pass otherwise); ws as for ssg_source_rowmin_filtered with the split copies. */
int ssg_source_rowmin_filtered1(const float* tgt, const float* src, int nrows, int Ns, int Ns_pad, int d, float tol, float scale_t,
                                float scale_s, float* ws, uint32_t* rowmin, ssg_stream_t stream);
/* v = half(1-exp(-rowmin)); *max_bits = max(v); v /= max(v)   (rerank.py:38-40).  A zero
 * max means the reference would produce NaNs (0/0): the caller must raise. */
int ssg_source_vec_finish(const uint32_t* rowmin, int N, uint16_t* v, uint32_t* max_bits, ssg_stream_t stream);

/* ---- K5 ranking (replaces np.argsort in reid/rerank.py:68-70) --------------------------- */
/* rank[il, 0:K] = the K smallest of half(D[il,:]/rowmax[il]) in (value, column) order
 * (== np.argsort(kind='stable'); opt-in rank_mode='stable'). K <= 64 */
int ssg_topk_rank(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, ssg_stream_t stream);
/* The same K columns in the order of the UNMODIFIED reference: np.argsort's default kind (rerank.py:70) is numpy's unstable
 * introsort on an index array (npysort aquicksort_<half>: median-of-3 Hoare partition, insertion sort below 17 entries,
 * heapsort past the depth budget), so the column of equal keys depends on the whole partition sequence.  One workgroup per
 * row replays exactly the partitions that reach columns [0,K) (csrc/topk_intro.hip).  2 <= N <= 131072, K <= 64.
 * Two launches: a workgroup per row runs the partitions of ranges longer than 2048 entries (SSG_INTRO_TAILN; 0 = one launch,
 * the round-2 kernel) and hands the shorter ranges that still intersect [0, K) to a one-wave-per-row tail kernel through ws.
 * ws: ssg_topk_rank_introsort_ws_bytes(N, nrows) bytes = the hand-over records (about 9 KB per row) plus, for rows that do not fit
 * in LDS (N > ~36 k), the global arena.  A caller that passes ssg_topk_rank_introsort_arena_bytes(N, nrows) bytes selects the
 * global-arena variant for any N (parity tests).  ws == NULL / fewer bytes than the hand-over records (the pre-round-3 calling
 * convention for LDS-resident rows) is accepted: the replay then runs unsplit in one launch (slower); only the arena is mandatory. */
size_t ssg_topk_rank_introsort_ws_bytes(int N, int nrows);
/* round 6 (test / diagnostic surface): byte offset, inside a workspace of ssg_topk_rank_introsort_arena_bytes(N, nrows) bytes, of the per-row
 * int32 flags of the streamed replay (1 = the row was handed to the in-place kernel: a pivot landed inside [0, K)); (size_t)-1 when the
 * streamed kernel does not run for this N.  Valid after the call that used the workspace. */
size_t ssg_topk_rank_introsort_flags_offset(int N, int nrows);
size_t ssg_topk_rank_introsort_arena_bytes(int N, int nrows);
int ssg_topk_rank_introsort(const uint16_t* D, const uint32_t* rowmax, int N, int nrows, int K, int32_t* rank, void* ws, size_t ws_bytes,
                            ssg_stream_t stream);

/* ---- K6 k-reciprocal encoding (reid/rerank.py:74-92) ------------------------------------ */
int ssg_krecip_row_capacity(int k1); /* entries per sparse V row: (k1+1)*(round(k1/2)+2) */
/* rank is the FULL [N,K] table; D/rowmax and the outputs cover rows [row0,row0+nrows).
 * Output: sparse rows sorted by column: v_idx/v_val [nrows,cap], v_nnz [nrows]. */
int ssg_krecip(const uint16_t* D, const uint32_t* rowmax, const int32_t* rank, int N, int row0, int nrows, int K, int k1, int cap,
               int32_t* v_idx, uint16_t* v_val, int32_t* v_nnz, ssg_stream_t stream);

/* ---- K7 local query expansion (reid/rerank.py:94-99) ------------------------------------ */
/* v_* are FULL [N,capV] tables; max_nnz = an upper bound of the V rows the k2 neighbours of these rows have (sizes the LDS staging:
 * max(v_nnz) read back, or a GUESS); q_* cover rows [row0,row0+nrows) with row stride capQ >= k2*max_nnz.
 * overflow (device int32 [2], zeroed by the caller, may be NULL; round 4): [0] = the longest V row met when one exceeds max_nnz (that
 * row of q_* is then truncated, never written out of bounds), [1] = the longest V row met at all -- lets the caller run on a guessed
 * max_nnz without a host round trip, redo the rare miss and size its next guess. */
int ssg_query_expand(const int32_t* v_idx, const uint16_t* v_val, const int32_t* v_nnz, const int32_t* rank, int N, int row0, int nrows,
                     int K, int k2, int capV, int capQ, int max_nnz, int32_t* q_idx, uint16_t* q_val, int32_t* q_nnz, int32_t* overflow,
                     ssg_stream_t stream);

/* ---- K8 inverted index (reid/rerank.py:101-103) ------------------------------------------ */
/* colcnt [ncols] int32 scratch, colptr [ncols+1] int64, inv_row/inv_val >= sum(q_nnz) entries */
int ssg_invert_index(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int nrows, int ncols, int capQ, int32_t* colcnt,
                     int64_t* colptr, int32_t* inv_row, uint16_t* inv_val, ssg_stream_t stream);

/* ---- K9 Jaccard distance (reid/rerank.py:105-122) ---------------------------------------- */
/* Jp[il,k] = half(clamp(1 - t/(2-t)) * half(1-lambda)); q_* are FULL tables; inv_nnz = colptr[N];
 * colmeta = caller workspace of 2*nrows*capQ int32. */
int ssg_jaccard_rows(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int capQ, const int64_t* colptr,
                     const int32_t* inv_row, const uint16_t* inv_val, int64_t inv_nnz, int32_t* colmeta, int N, int row0, int nrows,
                     uint16_t one_minus_lambda_half, uint16_t* Jp, ssg_stream_t stream);
/* Second generation (round 4): the same J' rows with every line written once, plus a SPARSE copy S of the columns the row's walk
 * touched -- every other column holds the constant J'(0) = half(1 - lambda), the largest value of the row.  s_pool [s_cap] uint32 =
 * packed (J' << 17 | column); seg_off / seg_len [nrows * ssg_jaccard_segments(N)] = the segment of every (row, 32768-column chunk);
 * s_cursor [2] uint64 (zeroed by the call): entries allocated, and 1 when a segment did not fit (S unusable: consumers go dense).
 * s_pool == NULL: J' only.  ssg_eps_compact_below_s / ssg_region_query_s are the passes that walk S instead of the N x N matrix. */
int ssg_jaccard_segments(int N);
int ssg_jaccard_rows2(const int32_t* q_idx, const uint16_t* q_val, const int32_t* q_nnz, int capQ, const int64_t* colptr,
                      const int32_t* inv_row, const uint16_t* inv_val, int64_t inv_nnz, int32_t* colmeta, int N, int row0, int nrows,
                      uint16_t one_minus_lambda_half, uint16_t* Jp, uint32_t* s_pool, uint64_t s_cap, uint64_t* s_cursor, int64_t* seg_off,
                      int32_t* seg_len, ssg_stream_t stream);
/* API materialisation of final_dist (rerank.py:122): out[il,k] = f64(Jp) + f64(half(v_i+v_k))*lambda */
int ssg_final_dist_f64(const uint16_t* Jp, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double* out,
                       ssg_stream_t stream);

/* ---- K10 epsilon rule (selftraining.py:289-293) ------------------------------------------ */
/* Matrix view for K10/K11 (M is [nrows,N] row-major): mode 0 = half Jp + v + lambda -> the
 * final_dist values of rerank.py:122; mode 1 = plain half matrix (no-rerank euclidean_dist);
 * mode 2 = plain float64 matrix (any precomputed distance, the sklearn drop-in case). */
/* One radix level over the strict upper triangle, zeros dropped: hist[bin] (uint64[4097]) +=
 * count of keys whose bits above (shift+width) equal prefix; hist[4096] += non-zero count. */
int ssg_eps_hist(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, uint64_t prefix,
                 int shift, int width, int count_nonzero, uint64_t* hist, ssg_stream_t stream);
/* buf[cursor++] = key for every strict-upper non-zero key <= key_max (writes beyond cap dropped) */
int ssg_eps_compact(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, uint64_t key_max,
                    uint64_t* buf, uint64_t cap, uint64_t* cursor, ssg_stream_t stream);
/* Fast path of the epsilon rule: (1) float32-surrogate histogram (4096 bins over the float bit pattern, hist[4096] = sample size;
 * hist = 4097 uint64 zeroed by the caller) of every row_stride-th local row's strict-upper non-zero elements; (2) threshold =
 * upper edge of the bin after the one where the cumulative count reaches quantile * sample size -> thr3 = {float bits, sample
 * size, bin}; (3) ONE pass over the block: exact float64 keys of the strict-upper non-zero elements whose surrogate is below the
 * threshold -> buf (cursor2[0] counts them, writes beyond cap dropped), cursor2[1] += exact zeros.  The caller sorts, and accepts
 * the result only if the top-th smallest key is below the threshold by more than the surrogate error (else: the radix select). */
int ssg_eps_sample_hist(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, int row_stride,
                        const uint64_t* refine, uint64_t* hist, ssg_stream_t stream);
/* thr = 5 uint64: {threshold float bits, sample size, selected coarse bin, sample elements below that bin, target rank} */
/* round 6: both sampling levels AND their selections in two launches (instead of four): the workgroup that finishes a level last selects.
 * hist2x = 2 x 4097 words and tickets2 = 2 words, zeroed by the caller; thr5 (5 words) ends up as ssg_eps_select_threshold followed by
 * ssg_eps_refine_threshold leave it (the separate calls remain: the sharded two-call form all-reduces the histograms in between) */
int ssg_eps_sample_threshold(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, int row_stride,
                             double quantile, uint64_t* hist2x, uint64_t* thr5, uint32_t* tickets2, uint64_t* splitters1023, ssg_stream_t stream);
/* splitters1023 (nullable): 1023 ascending float64 bit patterns that cut the values below the threshold into 1024 parts of about equal sample
 * mass (interpolated in the level-1 histogram) -- the splitters of ssg_samplesort_u64_presplit_dev for the keys the compaction pass collects */
int ssg_eps_select_threshold(const uint64_t* hist, double quantile, uint64_t* thr3, ssg_stream_t stream);
/* second level: ssg_eps_sample_hist(..., refine = thr, hist2) counts the sample elements of the selected coarse bin in 1024 linear
 * sub-bins (hist2 zeroed by the caller); this replaces thr[0] by the sub-bin edge (one guard sub-bin) */
int ssg_eps_refine_threshold(const uint64_t* hist2, uint64_t* thr, ssg_stream_t stream);
int ssg_eps_compact_below(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value,
                          const uint64_t* thr3, uint64_t* buf, uint64_t cap, uint64_t* cursor2, ssg_stream_t stream);
/* the same pass through the sparse copy S (mode 0 handles only): cursor3 = {keys collected, exact zeros, dense pass needed} zeroed by
 * the caller, vmin = ssg_half_min(v), rowmask = nrows bytes of workspace.  Row i is walked through S when its floor
 * J'(0) + lambda * half(v_i + vmin) -- a lower bound of every column outside S -- lies at or above the threshold; the other rows (all
 * of them when S overflowed or lambda < 0) are flagged in rowmask and done by the dense pass queued behind (inside this call), gated on
 * cursor3[2] -- decided on the device, no read-back in between */
int ssg_eps_compact_below_s(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, const uint64_t* thr3,
                            uint64_t* buf, uint64_t cap, uint64_t* cursor3, const uint32_t* s_pool, const int64_t* seg_off,
                            const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin, uint16_t jp0_half,
                            uint8_t* rowmask, ssg_stream_t stream);
/* *out_bits = the smallest of N non-negative halves (the source vector v of rerank.py:38-40) as half bits */
int ssg_half_min(const uint16_t* v, int N, uint32_t* out_bits, ssg_stream_t stream);
int ssg_fill_u64(uint64_t* buf, uint64_t n0, uint64_t n1, uint64_t value, ssg_stream_t stream);
int ssg_sort_u64(uint64_t* buf, uint64_t n_pow2, ssg_stream_t stream); /* ascending, n = 2^k >= 2048 */
/* round 5: the same sort with the number of keys left on the device (*n_dev, the compaction pass's cursor word): the network is launched
 * for the capacity n_cap (2^k >= 2048) but works on the power of two >= max(*n_dev, 2048) only; [*n_dev, that) is filled with ~0 first */
int ssg_sort_u64_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, ssg_stream_t stream);
/* round 5: the a-posteriori checks of the sampled eps rule on the device (selftraining.py:289-293 stays exact): top = rint(rho * (upper_total -
 * cursor[1])) == top_guess (the tree ssg_eps_mean_run summed), cursor[0] <= n_cap keys collected, >= top of them, the top-th sorted key below
 * the float32 threshold thr3[0] by its margin.  status6 = {ok, got, zeros, top, top-th key bits, threshold bits}; on failure eps2[0] := NaN */
int ssg_eps_check(const uint64_t* sorted_keys, const uint64_t* cursor, const uint64_t* thr3, double rho, uint64_t upper_total, int64_t top_guess,
                  uint64_t n_cap, double* eps2, uint64_t* status6, const uint64_t* sort_fail, ssg_stream_t stream);   /* sort_fail: nullable, ssg_samplesort_u64_dev's word */
/* round 5: the same device-sized ascending sort in 5 launches instead of 25 (sample sort: 1023 splitters out of a sorted sample of 4096 keys,
 * buckets ranked in LDS; keys equal to a splitter get their own bucket, so duplicates cost nothing).  In place through ws (any n_cap >= 1);
 * *fail = 1 when a bucket between two splitters holds more than 16384 keys (buf is then a permutation of the keys, not sorted) */
size_t ssg_samplesort_u64_workspace_bytes(uint64_t n_cap);
int ssg_samplesort_u64_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, void* ws, size_t ws_bytes, uint64_t* fail, ssg_stream_t stream);
/* round 6: a second geometry of the same sort for 4e5 .. 2.4e7 expected keys (4095 splitters out of a sorted sample of 16 384 keys, sorting
 * buckets of up to 16 384 keys in 128 KB of LDS, 1024 threads): N = 128 000 collects 1.7e7 candidates, where the bitonic network over the
 * whole array took 7.1 ms.  Same contract as ssg_samplesort_u64_dev (*fail = 1: a bucket beyond 16 384 keys), its own workspace size. */
size_t ssg_samplesort_u64_big_workspace_bytes(uint64_t n_cap);
int ssg_samplesort_u64_big_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, void* ws, size_t ws_bytes, uint64_t* fail, ssg_stream_t stream);
/* round 6: the same sort on 1023 ascending splitters the caller already holds on the device (any monotone splitters sort correctly; balance is
 * the caller's business: a bucket beyond 16 384 keys sets *fail): three launches, no sample sort.  gcount2048 (2048 uint32) and *fail: zero on entry */
int ssg_samplesort_u64_presplit_dev(uint64_t* buf, uint64_t n_cap, const uint64_t* n_dev, const uint64_t* splitters1023, uint32_t* gcount2048,
                                    void* ws, size_t ws_bytes, uint64_t* fail, ssg_stream_t stream);
size_t ssg_eps_mean_workspace_bytes(int64_t top);
/* out2[0] = mean of the first `top` sorted keys with numpy's pairwise summation (mode 0: f64;
 * mode 1: float32 sum of half values -> half, out2[1] = its bits) */
int ssg_eps_mean(const uint64_t* sorted_keys, int64_t top, int mode, void* ws, size_t ws_bytes, double* out2, ssg_stream_t stream);
/* the same in two steps: _prepare uploads the recursion tables of numpy's pairwise tree for `top` summands (blocks until the copy is
 * done: call it while the stream is idle), _run launches the summation asynchronously.  ssg_eps_mean = prepare + run. */
int ssg_eps_mean_prepare(int64_t top, void* ws, size_t ws_bytes, ssg_stream_t stream);
int ssg_eps_mean_run(const uint64_t* sorted_keys, int64_t top, int mode, void* ws, size_t ws_bytes, double* out2, ssg_stream_t stream);
/* round 6: ssg_eps_mean_run followed by ssg_eps_check without the check's own launch (same arguments as the two calls; eps2 = out2) */
int ssg_eps_mean_check(const uint64_t* sorted_keys, int64_t top_guess, int mode, void* ws, size_t ws_bytes, double* eps2, const uint64_t* cursor,
                       const uint64_t* thr3, double rho, uint64_t upper_total, uint64_t n_cap, uint64_t* status6, const uint64_t* sort_fail, ssg_stream_t stream);

/* ---- K11/K12 DBSCAN (selftraining.py:295,306; sklearn 1.7.2 DBSCAN precomputed) ---------- */
/* cnt[il] = |{k: d(i,k) <= eps}|; edges[2e],[2e+1] = (i,k) for every hit (cursor counts all) */
int ssg_region_query(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, double eps,
                     int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor, ssg_stream_t stream);
/* the same through S, row by row: row i through S when J'(0) + lambda * half(v_i + vmin) > eps, the dense scan (queued behind, gated on
 * cursor2[1]) for the rows flagged in rowmask.  cursor2 = {edges counted, dense pass needed}, zeroed by the caller. */
int ssg_region_query_s(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, double eps, const uint32_t* s_pool,
                       const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                       uint16_t jp0_half, uint8_t* rowmask, int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor2,
                       ssg_stream_t stream);
/* round 5: both region queries with eps read from device memory (*eps_dev, e.g. out2[0] of ssg_eps_mean_run after ssg_eps_check): the eps rule,
 * the region query and the components run back to back, the host reads eps with the labels.  *eps_dev = NaN: no hit */
int ssg_region_query_dev(const void* M, const uint16_t* v, int N, int row0, int nrows, int mode, double lambda_value, const double* eps_dev,
                         int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor, ssg_stream_t stream);
int ssg_region_query_s_dev(const void* M, const uint16_t* v, int N, int row0, int nrows, double lambda_value, const double* eps_dev, const uint32_t* s_pool,
                           const int64_t* seg_off, const int32_t* seg_len, int nseg, const uint64_t* s_cursor, const uint32_t* vmin,
                           uint16_t jp0_half, uint8_t* rowmask, int32_t* cnt, int32_t* edges, uint64_t cap_edges, uint64_t* cursor2,
                           ssg_stream_t stream);
/* round 6 (sharded eps rule + DBSCAN as one device chain): nseg fixed-capacity segments of 8-byte items -- the all-gathered
 * candidate-key or edge buffers of the ranks; segment s holds min(counts[s * count_stride], seg_cap) valid items at in + s * seg_stride -- are
 * written to `out` back to back in segment order without the host seeing the counts.  total2[0] = items written, total2[1] = 1 when a
 * segment's count exceeded seg_cap (its owner's buffer overflowed).  out holds nseg * seg_cap items. */
int ssg_concat_segments_u64(const uint64_t* in, int nseg, uint64_t seg_cap, uint64_t seg_stride, const uint64_t* counts, int count_stride,
                            uint64_t* out, uint64_t* total2, ssg_stream_t stream);
size_t ssg_dbscan_cc_workspace_bytes(int N);
/* cnt is the FULL [N] table, edges the concatenated edge list: labels[N] int64, -1 = noise */
int ssg_dbscan_cc(const int32_t* cnt, const int32_t* edges, uint64_t nedges, int N, int min_samples, void* ws, size_t ws_bytes,
                  int64_t* labels, ssg_stream_t stream);
/* the same with the edge count left on the device (the region query's cursor; min(*nedges_dev, cap_edges) edges are read): no host
 * round trip between region query and labels.  labels may be NULL (round 6): the int64 conversion launch is skipped and the caller reads
 * the labels as int32 from the workspace itself -- lab = (int32_t*)ws + N, 0x7fffffff = noise (ws: parent[N] | lab[N] | ...) */
int ssg_dbscan_cc_dev(const int32_t* cnt, const int32_t* edges, const uint64_t* nedges_dev, uint64_t cap_edges, int N, int min_samples,
                      void* ws, size_t ws_bytes, int64_t* labels, ssg_stream_t stream);

/* ---- K1/K2 ResNet-50 embedding forward (reid/models/resnet.py:86-111, reid/evaluators.py:18-60) */
/* Conv2d with eval-BatchNorm folded into (w, bias) + optional residual add + optional ReLU, NHWC
 * float32 on the fp32 matrix cores.  in [B,H,W,Cin]; w [Cout][Kpad], k = ((c/32)*KH*KW + r*KW+s)*32 + c%32
 * (32-channel chunks outermost: the taps of a chunk re-read cached pixels; stem: k = (r*KW+s)*4 + c), rows
 * zero-padded to Kpad (multiple of 32); res/out [B,OH,OW,Cout].  Cin % 32 == 0 or Cin == 4 (stem:
 * RGB0 pixels, Kpad = 32*ceil(KH*KW/8)); Cout % 64 == 0.  (cuDNN conv+BN+ReLU of base.py:57-93) */
int ssg_conv2d_nhwc_f32(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, int relu, ssg_stream_t stream);
/* Bottleneck tail with a downsample branch (base.py:75-90) as one GEMM over the concatenated K:
 * out = relu(conv3(in) + downsample(in2) + bias); w [Cout][Cin+Cin2], bias = folded b3 + b_ds;
 * in [B,H,W,Cin] (1x1 stride 1), in2 [B,H2,W2,Cin2] sampled at (oh*stride2, ow*stride2). */
int ssg_conv1x1_dual_nhwc_f32(const float* in, const float* in2, const float* w, const float* bias, float* out, int B, int H, int W, int Cin,
                              int H2, int W2, int Cin2, int stride2, int Cout, int relu, ssg_stream_t stream);
/* Same two layers with the fp32 values carried as SPLIT HALVES ("h8l8": per 8 consecutive channels 32 bytes =
 * [8 x half hi][8 x half lo], hi = half(v), lo = half(v - hi); same 4 bytes per value and same addressing as
 * fp32).  flags & SSG_CONV_IN_SPLIT: in (in2) and w are h8l8, w pre-multiplied by 1/acc_scale (a power of two
 * that lifts small weights out of the half subnormals); the GEMM then runs on v_mfma_f32_32x32x16_f16 as
 * xh*wh + xh*wl + xl*wh with fp32 accumulation (half x half products are exact in fp32, the dropped xl*wl is
 * < 2^-22 |x*w|): fp32-class results at 3/16 of the fp32-MFMA cost.  flags & SSG_CONV_OUT_SPLIT: out and res
 * are h8l8 (values must stay below 65504).  flags = 0 is exactly ssg_conv2d_nhwc_f32. */
#define SSG_CONV_IN_SPLIT 1
#define SSG_CONV_OUT_SPLIT 2
/* ch_scale (nullable): per-output-channel factor applied to the accumulator (after acc_scale) before the bias -- the split path
 * pre-multiplies every weight ROW by its own power of two, so folded checkpoints whose per-channel BN scales span orders of
 * magnitude keep all their bits.  overflow (nullable): *overflow is set to 1 when a value that does not fit the split-half
 * output format (|v| >= 65520 or NaN) is written; the caller zeroes it, reads it after the forward and falls back to fp32. */
int ssg_conv2d_nhwc_x(const void* in, const void* w, const float* bias, const void* res, void* out, int B, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride, int pad, int relu, int flags, float acc_scale, const float* ch_scale,
                      int32_t* overflow, ssg_stream_t stream);
int ssg_conv1x1_dual_nhwc_x(const void* in, const void* in2, const void* w, const float* bias, void* out, int B, int H, int W, int Cin,
                            int H2, int W2, int Cin2, int stride2, int Cout, int relu, int flags, float acc_scale, const float* ch_scale,
                            int32_t* overflow, ssg_stream_t stream);
/* Round 6, experimental (opt-in through SSG_CONV_PAIR=1 in the Python layer): the tail of one identity bottleneck block and the head of the
 * next as ONE launch -- out [M,C] = relu(conv3_1x1(y2 [M,K1]) + b3 + res [M,C]) (base.py:84-90) and y1n [M,N2] = relu(conv1_1x1(out) + b1n)
 * (base.py:76-78 of the next block); every tensor h8l8, weights as ssg_conv2d_nhwc_x takes them (w3 [C][K1], w1n [N2][C], rows
 * pre-multiplied by the powers of two cs3 / cs1n undo).  Bit-identical to the two ssg_conv2d_nhwc_x launches it replaces; out is still
 * written (the next block's residual) but not read back.  ssg_conv_pair_supported: 1 for the layer3 shape (K1 256, C 1024, N2 256). */
int ssg_conv_pair_supported(int K1, int C, int N2);
int ssg_conv_pair_nhwc_x(const void* y2, const void* w3, const float* b3, const float* cs3, const void* res, void* out,
                         const void* w1n, const float* b1n, const float* cs1n, void* y1n, int M, int K1, int C, int N2,
                         int32_t* overflow, ssg_stream_t stream);
/* One identity bottleneck block (base.py:57-90 without a downsample branch, stride 1) in ONE launch, split-half tensors:
 * out = relu(conv3(relu(conv2_3x3(relu(conv1(x))))) + x).  x / out [B,H,W,C] h8l8 (out must not alias x); w1 [MID][C],
 * w2 [MID][9*MID] (k = (32-channel chunk, tap, channel)), w3 [C][MID] as ssg_conv2d_nhwc_x takes them, rows pre-multiplied by
 * powers of two that cs1/cs2/cs3 undo; b* fp32 folded BatchNorm biases.  Bit-identical to the three ssg_conv2d_nhwc_x launches;
 * the two MID-channel intermediates stay in LDS.  ssg_bottleneck_supported() tells which block shapes have a kernel
 * (layer1 of ResNet-50 at 256x128 input: H x 32 x 256, MID 64; CIN == C for the identity block, CIN = 64 for the first block).
 * ssg_bottleneck_ds_nhwc_x: the block with a stride-1 downsample branch, out = relu(conv3(...) + downsample(x)); x [B,H,W,CIN],
 * w3cat [C][MID + CIN] = conv3 | downsample weights along K and b3 = b3 + b_ds, as ssg_conv1x1_dual_nhwc_x takes them. */
int ssg_bottleneck_supported(int H, int W, int CIN, int C, int MID);
int ssg_bottleneck_nhwc_x(const void* x, const void* w1, const float* b1, const float* cs1, const void* w2, const float* b2, const float* cs2,
                          const void* w3, const float* b3, const float* cs3, void* out, int B, int H, int W, int C, int MID,
                          int32_t* overflow, ssg_stream_t stream);
int ssg_bottleneck_ds_nhwc_x(const void* x, const void* w1, const float* b1, const float* cs1, const void* w2, const float* b2, const float* cs2,
                             const void* w3cat, const float* b3, const float* cs3, void* out, int B, int H, int W, int CIN, int C, int MID,
                             int32_t* overflow, ssg_stream_t stream);
/* The stem in ONE launch (base.py:101-105 conv1 + bn1 + relu + maxpool, with the fliplr of evaluators.py:12-16 folded into the
 * image read): images [B,3,H,W] float32 NCHW -> out [B,H/4,W/4,64] h8l8.  w [64][224] / bias / ch_scale: the stem weights as
 * ssg_conv2d_nhwc_x takes them (Cin = 4 "h4l4" layout).  Bit-identical to ssg_nchw_to_nhwc4_h4l4 + ssg_conv2d_nhwc_x +
 * ssg_maxpool3x3s2_h8l8; the 64-channel stem map never reaches HBM.  ssg_stem_pool_supported(): W == 128, H % 4 == 0. */
int ssg_stem_pool_supported(int H, int W);
int ssg_stem_pool_nchw_x(const float* images, int flip, const void* w, const float* bias, const float* ch_scale, void* out, int B, int H, int W,
                         int32_t* overflow, ssg_stream_t stream);
/* stem input for the split path: [B,3,H,W] NCHW fp32 -> [B,H,W] pixels of 16 bytes [4 x half hi][4 x half lo] ("h4l4",
 * 4th channel 0); ssg_conv2d_nhwc_x with Cin = 4 and SSG_CONV_IN_SPLIT takes these, with w in the same per-tap layout */
int ssg_nchw_to_nhwc4_h4l4(const float* in, void* out, int B, int H, int W, int flip, ssg_stream_t stream);
/* MaxPool2d(3,2,1) and the global/stripe average pool on h8l8 maps (the averages come out as fp32) */
int ssg_maxpool3x3s2_h8l8(const void* in, void* out, int B, int H, int W, int C, ssg_stream_t stream);
int ssg_gap_stripes_h8l8(const void* in, float* out, int B, int H, int W, int C, int num_split, ssg_stream_t stream);
/* fp32 [n] -> h8l8 of (in * scale), and h8l8 -> fp32 (times scale); n % 8 == 0 */
int ssg_h8l8_encode(const float* in, void* out, int64_t n, float scale, ssg_stream_t stream);
int ssg_h8l8_decode(const void* in, float* out, int64_t n, float scale, ssg_stream_t stream);
/* float32 squared-L2 block (reid/evaluators.py:63-85 pairwise_distance) on the fp32 matrix cores:
 * out[i,j] = |x_i|^2 + |y_j|^2 - 2<x_i,y_j>; self_form != 0 gives the reference's query=None form
 * 2|x_i|^2 - 2<x_i,y_j> (:64-72).  x [m,d], y [n,d], out [m,n]; d % 32 == 0, n % 64 == 0; ws = m+n floats. */
int ssg_pairwise_sqdist_f32(const float* x, const float* y, int m, int n, int d, int self_form, float* ws, float* out, ssg_stream_t stream);
/* [B,3,H,W] NCHW -> [B,H,W,4] NHWC (4th channel 0); flip != 0 mirrors W (evaluators.py:12-16 fliplr) */
int ssg_nchw_to_nhwc4(const float* in, float* out, int B, int H, int W, int flip, ssg_stream_t stream);
/* MaxPool2d(3, stride 2, padding 1) on NHWC (base.py:105) */
int ssg_maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, ssg_stream_t stream);
/* out[s][b][c]: s=0 global average pool, s=1..S the S horizontal stripes (resnet.py:93-111) */
int ssg_gap_stripes(const float* in, float* out, int B, int H, int W, int C, int num_split, ssg_stream_t stream);
/* out = (a+b)/||a+b||_2 per row (evaluators.py:31-35: original + flipped features, L2 norm) */
int ssg_flip_sum_l2norm(const float* a, const float* b, float* out, int rows, int C, ssg_stream_t stream);

/* ---- kNN-set Jaccard re-ranking variant (reid/rerank_plain.py:125-178 re_ranking; shares K3/K4 with rerank.py) */
/* A_i = { j != i : D[i,j] <= k-th smallest of row i } (:165-170).  rank = ssg_topk_rank of the same rows with rowmax = half(1)
 * everywhere and K = k; a_idx/a_val [nrows, cap] (a_val = half 1: feed ssg_invert_index), a_nnz [nrows]; *overflow = rows
 * with more than cap members. */
int ssg_knn_sets(const uint16_t* D, const int32_t* rank, int N, int row0, int nrows, int K, int cap, int32_t* a_idx, uint16_t* a_val,
                 int32_t* a_nnz, int32_t* overflow, ssg_stream_t stream);
/* J'[i,k] = half(half(|A_i xor A_k| / |A_i or A_k|) * half(1-lambda)) (scipy cdist 'jaccard' on booleans, :173-175; 0 for
 * two empty sets) for rows [row0,row0+nrows); a_idx/a_nnz cover all N rows, colptr/inv_row = ssg_invert_index of them. */
int ssg_set_jaccard_rows(const int32_t* a_idx, const int32_t* a_nnz, int capA, const int64_t* colptr, const int32_t* inv_row, int N, int row0,
                         int nrows, uint16_t one_minus_lambda_half, uint16_t* Jp, ssg_stream_t stream);

/* ---- retrieval metrics of the evaluation step (reid/evaluators.py:88-129 evaluate_all ->
 * reid/evaluation_metrics/ranking.py:18-79 cmc, :82-115 mean_ap + sklearn average_precision_score) */
/* dist [m, ld] float32 query x gallery block; ids / cams int32.  first_rank[q] = number of valid gallery entries
 * (different id or different camera; separate_cams != 0: different camera only, ranking.py:49-51) ordered before
 * the first true match of query q in (distance, gallery index) order, -1 when q has no valid true match;
 * ap[q] = its average precision (step-wise over distinct match distances like sklearn; NaN likewise).
 * *overflow = number of queries with more than 2048 true matches (not evaluated). */
int ssg_rank_metrics(const float* dist, int m, int n, int64_t ld, const int32_t* qid, const int32_t* qcam, const int32_t* gid,
                     const int32_t* gcam, int separate_cams, int32_t* first_rank, double* ap, int32_t* overflow, ssg_stream_t stream);
/* the same plus the 'allshots' CMC bins (ranking.py:62-75 with first_match_break=False): nmatch[q] = valid true matches of query q,
 * nm_before[q, s] (row stride nm_cap, s < min(nmatch[q], nm_cap)) = valid non-matching gallery entries ordered before its s-th match */
int ssg_rank_metrics_all(const float* dist, int m, int n, int64_t ld, const int32_t* qid, const int32_t* qcam, const int32_t* gid,
                         const int32_t* gcam, int separate_cams, int32_t* first_rank, double* ap, int32_t* overflow, int32_t* nm_before,
                         int32_t* nmatch, int nm_cap, ssg_stream_t stream);

/* ---- float32 re-ranking variant "re_ranking_init" (reid/rerank.py:171-234 == reid/rerank_initial.py:40-99) */
/* out[i,j] = 2 - 2<x_i,y_j> (rerank.py:174-182); d % 32 == 0, n % 64 == 0; zeros = n floats of 0 */
int ssg_cosine_dist_f32(const float* x, const float* y, int m, int n, int d, const float* zeros, float* out, ssg_stream_t stream);
int ssg_affine_2m2x_f32(const float* in, float* out, int64_t n, ssg_stream_t stream); /* out = 2 - 2*in (rerank_initial.py:50) */
/* D [N,N] symmetric float32: row max, top-(k1+1) ranking, k-reciprocal encoding -> sparse V (rerank.py:183-204) */
int ssg_rerank_init_stage1(const float* D, int N, int k1, int k2, int capV, float* rowmax, int32_t* rank, int32_t* v_idx, float* v_val,
                           int32_t* v_nnz, ssg_stream_t stream);
/* query expansion V_qe (rerank.py:207-212); max_nnz = max(v_nnz), capQ >= k2*max_nnz */
int ssg_rerank_init_expand(const int32_t* v_idx, const float* v_val, const int32_t* v_nnz, const int32_t* rank, int N, int k1, int k2, int capV,
                           int capQ, int max_nnz, int32_t* q_idx, float* q_val, int32_t* q_nnz, ssg_stream_t stream);
/* inverted index + Jaccard + blend for the nq query rows -> out [nq, N-nq] (rerank.py:214-233) */
int ssg_rerank_init_jaccard(const float* D, const float* rowmax, const int32_t* q_idx, const float* q_val, const int32_t* q_nnz, int capQ, int N,
                            int nq, float lambda_value, int32_t* colcnt, int64_t* colptr, int32_t* inv_row, float* inv_val, float* out,
                            ssg_stream_t stream);

/* ---- input transform of the extraction loaders (selftraining.py:43-47 applied by reid/utils/data/preprocessor.py:22-30):
 * Resize((H,W)) [= PIL.Image.resize((W,H), BILINEAR) on 8-bit RGB] + ToTensor + Normalize for a batch of equally sized decoded
 * images.  src [B,h,w,3] uint8, tmp [B,h,W,3] uint8 scratch, out [B,3,H,W] float32.  (xmin, xcnt, xk[W,xksize]) and
 * (ymin, ycnt, yk[H,yksize]) are Pillow's per-output-pixel windows of 22-bit fixed-point triangle coefficients
 * (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc; ssg_amd/preprocessor.py computes them), device arrays;
 * mean3_host / std3_host are HOST pointers to 3 floats.  Bit-exact with Pillow 12.2. */
int ssg_preprocess_u8(const uint8_t* src, int B, int h, int w, int H, int W, const int32_t* xmin, const int32_t* xcnt, const int32_t* xk,
                      int xksize, const int32_t* ymin, const int32_t* ycnt, const int32_t* yk, int yksize, const float* mean3_host,
                      const float* std3_host, uint8_t* tmp, float* out, ssg_stream_t stream);
/* Decode half of reid/utils/data/preprocessor.py:22-30 (`Image.open(fpath).convert('RGB')` = Pillow -> libjpeg-turbo, default
 * parameters) for a batch of baseline / extended-sequential Huffman JPEG files, bit-exact: Huffman decode (one thread per restart
 * segment), dequantisation + islow integer inverse DCT, fancy chroma upsampling (h2v1 / h2v2), YCbCr -> RGB.  The host parses the
 * marker segments and builds the tables (ssg_amd/jpeg.py):
 *   ecs   entropy-coded bytes of all files back to back, >= 32 zero bytes of padding behind them
 *   segs  int64 [nseg][5] per restart segment: image, byte offset into ecs, byte length, first MCU, MCU count
 *   imgs  int64 [nimg][32] per image: W, H, components (1 | 3), luma sampling h, v, MCUs per row, MCU rows, byte offset into out, then
 *         8 words per component: first block in coef, blocks per row, block rows, byte offset into planes, plane pitch, quantisation
 *         table index, DC table index, AC table index
 *   look [ntab][256] uint16 ((length << 8) | symbol of every code of at most 8 bits, 0 otherwise), maxcode [ntab][18], valoff [ntab][17],
 *   vals [ntab][256]: jdhuff.c's derived tables; qts [nqt][64] uint16 in natural order
 *   coef  workspace int16 [total_blocks][64] (zeroed by the call), planes workspace uint8 (sum of 64 * blocks), max_blocks / max_pixels =
 *         largest component (in blocks) / image (in pixels) of the batch; out: RGB bytes, H * W * 3 per image at its offset.
 *   status int32 [nimg] (zeroed by the call; round 4): non-zero = damaged entropy-coded data (bit 0: a zero run past coefficient 63,
 *         bit 1: a segment ran out of data before its MCUs were decoded) -- the pixels of such a file are not libjpeg's; the caller
 *         hands it to the reference's decoder (Pillow), which warns / raises like the reference. */
int ssg_jpeg_decode_batch(const uint8_t* ecs, const int64_t* segs, int nseg, const int64_t* imgs, int nimg, const uint16_t* look,
                          const int32_t* maxcode, const int32_t* valoff, const uint8_t* vals, const uint16_t* qts, int16_t* coef,
                          int64_t total_blocks, int max_blocks, uint8_t* planes, int max_pixels, uint8_t* out, int32_t* status,
                          ssg_stream_t stream);
/* Host side of the same path, native and threaded (csrc/jpeg_host.hip; no device work): the marker walk + table building that
 * ssg_amd/jpeg.py states in Python, for a batch of files in host memory -- what the reference's DataLoader workers do in libjpeg's
 * jdmarker.c before decoding.  ssg_jpeg_parse_open: pass over files[i] (lens[i] bytes each; must stay valid until _close) on
 * `nthreads` threads; counts10 = [images the GPU decodes, restart segments, bytes of entropy-coded data incl. 64 of padding, Huffman
 * tables, quantisation tables, coefficient blocks, largest component (blocks), plane bytes, output bytes, largest image (pixels)];
 * file_status[i] = 0 (decoded on the GPU, images are numbered in file order) or 1 (not baseline / malformed header: left to the
 * reference's decoder).  ssg_jpeg_parse_fill writes the arguments of ssg_jpeg_decode_batch into host buffers of those sizes. */
int ssg_jpeg_parse_open(const void* const* files, const int64_t* lens, int nfiles, int nthreads, void** handle, int64_t* counts10,
                        int32_t* file_status);
int ssg_jpeg_parse_fill(void* handle, int64_t* imgs, int64_t* segs, uint8_t* pool, uint16_t* look, int32_t* maxcode, int32_t* valoff,
                        uint8_t* vals, uint16_t* qts);
int ssg_jpeg_parse_close(void* handle);
/* x = sqrt(max(x, lo)) in place: with ssg_pairwise_sqdist_f32 the pairwise block of the fine-tune phase's TripletLoss
 * (reid/loss/triplet.py:28-31: dist = (|x|^2 + |x|^2' - 2 x x').clamp(min=1e-12).sqrt()) */
int ssg_clamp_sqrt_f32(float* x, int64_t n, float lo, ssg_stream_t stream);
/* Backward of that block (round 4: lets ssg_amd.triplet.pairwise_dist replace reid/loss/triplet.py:28-31 inside the training step):
 * grad_x = diag(rowsum(S)) x - S x,  S = W + W^T,  W[i,j] = sq[i,j] >= lo ? grad_dist[i,j] / dist[i,j] : 0 (sq = the distances squared
 * before the clamp).  _weights writes S with row pitch ld >= n (zero padded) and its row sums; S x is one fp32-MFMA GEMM
 * (ssg_conv2d_nhwc_f32 as a 1x1 convolution: in = S [n,1,1,ld], w = x^T [d_pad][ld]); _combine forms the gradient. */
int ssg_triplet_grad_weights(const float* grad_dist, const float* sq, const float* dist, int n, int ld, float lo, float* S, float* rowsum,
                             ssg_stream_t stream);
int ssg_triplet_grad_combine(const float* x, const float* rowsum, const float* Sx, int n, int d, int ldo, float* grad_x, ssg_stream_t stream);

/* ---- device self-tests used by the parity suite ----------------------------------------- */
int ssg_selftest_half_table(int which, uint16_t* out65536, ssg_stream_t stream);
int ssg_selftest_half_binop(int which, const uint16_t* a, const uint16_t* b, int n, uint16_t* out, ssg_stream_t stream);
int ssg_selftest_d2h(const double* a, int n, uint16_t* out, ssg_stream_t stream);

/* ---- collectives of the sharded path: RCCL over xGMI, one rank per process / GPU ------------------------
 * The reference spreads the extraction over GPUs with nn.DataParallel (selftraining.py:135) and has no multi-GPU N x N path.
 * Every exchange of the sharded pipeline is an all-gather of equally sized blocks (embeddings, rank lists, sparse V / V_qe,
 * source vector) or an int64 sum all-reduce (eps histogram).  The Python product issues them through torch.distributed
 * (backend "nccl" = RCCL on ROCm); a host language without torch binds these four calls instead.
 * ssg_comm_unique_id: 128 host bytes created by rank 0, passed by EVERY rank to ssg_comm_init (a collective; the communicator
 * binds to the caller's current HIP device).  ssg_allgather: recv[r * bytes_per_rank ...] = rank r's block. */
/* RCCL is bound at the first collective call, to the build that is already mapped into the process if there is one (torch's
 * torch/lib/librccl.so next to torch.distributed), else $SSG_RCCL_PATH / librccl.so.1: never two RCCL builds in one process.
 * ssg_comm_library() = path of the bound library ("" if none). */
const char* ssg_comm_library(void);
int ssg_comm_unique_id(void* id128_host);
int ssg_comm_init(void** comm, int world, int rank, const void* id128_host);
int ssg_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, ssg_stream_t stream);
int ssg_allreduce_sum_i64(void* comm, int64_t* buf, size_t count, ssg_stream_t stream);
int ssg_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SSG_HIP_H */
