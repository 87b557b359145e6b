"""Per-launch HIP-event timing of the C-ABI entry points and the roofline objects built from it -- shared by bench.py (the headline
configuration) and tools/run_configs.py (BASELINE configs[3] / [4]), so that every size is reported in the same terms.

Peaks: /opt/skills/guides/MI355X_MICROARCH.md (HBM3E 8 TB/s; dense fp16 2516.6 TFLOP/s, int8 5033.2 TOP/s, fp64 78.6, fp32 157.3)."""
import torch

PEAK_FP32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_FP16_MFMA_TF = 2516.6    # dense fp16/bf16 matrix peak (v_mfma_f32_32x32x16_f16)
PEAK_INT8_MFMA_TOPS = 5033.2  # dense int8 matrix peak (v_mfma_i32_32x32x32_i8 = 2x the fp16 rate)
PEAK_FP64_MFMA_TF = 78.6      # SURVEY.md 8d / BASELINE.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec

K5_K12 = ("ssg_topk_rank", "ssg_topk_rank_introsort", "ssg_krecip", "ssg_query_expand", "ssg_invert_index", "ssg_jaccard_rows", "ssg_jaccard_rows2", "ssg_half_min",
          "ssg_eps_hist", "ssg_eps_compact", "ssg_eps_sample_hist", "ssg_eps_select_threshold", "ssg_eps_refine_threshold", "ssg_eps_compact_below",
          "ssg_eps_compact_below_s", "ssg_fill_u64", "ssg_sort_u64", "ssg_sort_u64_dev", "ssg_samplesort_u64_dev", "ssg_samplesort_u64_big_dev", "ssg_samplesort_u64_presplit_dev", "ssg_eps_sample_threshold", "ssg_eps_mean_check",
          "ssg_concat_segments_u64", "ssg_eps_check", "ssg_eps_mean", "ssg_eps_mean_run", "ssg_region_query", "ssg_region_query_s",
          "ssg_region_query_dev", "ssg_region_query_s_dev", "ssg_dbscan_cc",
          "ssg_dbscan_cc_dev")


class KernelTimer:
    """HIP-event timing of individual C-ABI launches on the stream they are launched on."""

    def __init__(self, L):
        self.L, self.ev, self.on = L, {}, False
        self.sample, self.sampled_images = True, 0      # embed batches are sampled 1 in 8 (the events cost 2.7 % when on every launch, and a sampled batch runs its two forwards on one stream)

    def __getattr__(self, k):
        fn = getattr(self.L, k)
        if not self.on or not self.sample or not k.startswith("ssg_") or k.endswith("_bytes") or k.endswith("_supported") or k in (
                "ssg_last_error", "ssg_krecip_row_capacity", "ssg_double_to_half_bits", "ssg_version", "ssg_eps_mean_prepare"):
            return fn

        def timed(*a):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*a); e1.record()
            self.ev.setdefault(k, []).append((e0, e1))
            return rc
        return timed

    def totals(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.ev.items()}



def grouping_roofline(tot, N, nrows, Ns, world, steps, d=2048, row_split=None):
    """tot = KernelTimer.totals() of `steps` grouping legs on a row block of `nrows` of N rows -> (roofline_kernels list, roofline_k5_k12 dict):
    every N x N streaming kernel against its own 2*N^2 (or N^2) algorithmic bytes, the matrix-core kernels against the peak of
    the instruction they execute, and K5..K12 together against 8*N^2 bytes.
    row_split: {'eps': (rows through the sparse copy, rows through the dense pass), 'region': (...)} from cluster.sparse_row_split (round 6): the
    sparse passes' objects then say which path the rows took, and a pass whose rows ALL went dense is listed as the streaming pass it was."""
    nn2 = 2.0 * nrows * N   # bytes of one half row block
    hbm = []
    for k, byt, what in (("ssg_topk_rank", nn2, "reads D (2*N^2 B); canonical (value, column) order, opt-in rank_mode='stable'"),
                         ("ssg_topk_rank_introsort", nn2, "reads D (2*N^2 B); replays numpy's unstable introsort argsort per row (reference tie order, "
                          "default): VALU/latency-bound emulation of a sequential algorithm, listed against the same bytes"),
                         ("ssg_jaccard_rows", nn2, "writes J' (2*N^2 B)"),
                         ("ssg_jaccard_rows2", nn2, "writes J' (2*N^2 B), every line once, + the sparse copy S of the touched columns (round 4)"),
                         ("ssg_eps_hist", nn2 / 2, "radix-select fallback of the eps rule: reads upper triangle of J' (N^2 B) per level"),
                         ("ssg_eps_compact", nn2 / 2, "radix-select fallback: reads upper triangle of J' (N^2 B)"),
                         ("ssg_eps_compact_below", nn2 / 2, "eps rule, the one full pass of the sampled-threshold path: reads upper triangle of J' (N^2 B)"),
                         ("ssg_eps_compact_below_s", nn2 / 2, "eps rule, the one full pass, through the sparse copy S of J' where the threshold lies below the row's floor "
                          "(dense scan of the other rows): listed against the N^2 B of the upper triangle it replaces"),
                         ("ssg_region_query", nn2, "reads J' (2*N^2 B)"),
                         ("ssg_region_query_dev", nn2, "reads J' (2*N^2 B); eps read from device memory"),
                         ("ssg_region_query_s_dev", nn2, "region query through the sparse copy S of J' where eps (read from device memory) lies below the row's floor "
                          "(dense scan of the other rows): listed against the 2*N^2 B it replaces"),
                         ("ssg_region_query_s", nn2, "region query through the sparse copy S of J' where eps lies below the row's floor (dense scan of the other rows): "
                          "listed against the 2*N^2 B it replaces")):
        if k in tot:
            n, ms = tot[k]
            gbs = byt * n / (ms * 1e-3) / 1e9
            split = (row_split or {}).get("eps" if "compact" in k else "region") if (k.endswith("_s") or k.endswith("_s_dev")) else None
            if split is not None and split[0] == 0 and split[1] > 0:
                # every row was flagged for the dense pass queued behind the sparse kernel: it WAS the streaming pass -- bandwidth and fraction apply
                hbm.append({"kernel": k, "bound": "hbm", "what": what + " -- ALL rows took the dense pass here", "launches": n, "avg_launch_ms": round(ms / n, 4),
                            "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "rows_sparse": 0, "rows_dense": int(split[1])})
                continue
            if k.endswith("_s") or k.endswith("_s_dev"):
                # the sparse passes do not stream the matrix: no bandwidth figure, no roofline fraction -- the bytes they no longer move
                hbm.append({"kernel": k, "bound": "latency (L2 round trips over a few hundred packed words + gathers of v per row)", "what": what, "launches": n,
                            "avg_launch_ms": round(ms / n, 4), "bytes_avoided": int(byt), "unit": "B per launch not streamed",
                            "dense_pass_time_at_hbm_peak_ms": round(byt / (PEAK_HBM_GBS * 1e9) * 1e3, 4), "frac": None,
                            "note": "walks the sparse copy S instead of the N x N matrix; `bytes_avoided` = the algorithmic bytes of the dense pass it replaces "
                                    "(SURVEY.md 8d); a roofline fraction of bytes that are not moved would be meaningless (it exceeded 1 in round 4)"})
                if split is not None:
                    hbm[-1]["rows_sparse"], hbm[-1]["rows_dense"] = int(split[0]), int(split[1])
                    hbm[-1]["bytes_avoided"] = int(byt * split[0] / max(split[0] + split[1], 1))
                continue
            hbm.append({"kernel": k, "bound": "hbm", "what": what, "launches": n, "avg_launch_ms": round(ms / n, 4), "achieved": round(gbs, 1),
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
    # the self term computes only the upper-triangle tiles when one GPU holds the whole matrix (mirrored on store)
    t128 = (N + 127) // 128
    self_flop = (t128 * (t128 + 1) // 2) * 128 * 128 * 2.0 * d if world == 1 else 2.0 * nrows * N * d
    for k, flop in (("ssg_sqdist_self_f16", self_flop), ("ssg_source_rowmin_f16", 2.0 * nrows * Ns * d)):
        if k in tot:
            n, ms = tot[k]
            tf = flop * n / (ms * 1e-3) / 1e12
            hbm.append({"kernel": k, "bound": "mfma", "what": "fp64 Gram (v_mfma_f64_16x16x4); executed flops (self term: upper-triangle tiles only on 1 GPU)", "launches": n, "avg_launch_ms": round(ms / n, 4),
                        "achieved": round(tf, 2), "peak": PEAK_FP64_MFMA_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP64_MFMA_TF, 4)})
    if "ssg_sqdist_self_i8" in tot:
        n, ms = tot["ssg_sqdist_self_i8"]
        t64 = (N + 63) // 64
        alg = ((t64 * (t64 + 1) // 2) * 64 * 64 if world == 1 else nrows * N) * 2.0 * d      # executed tiles (upper triangle on 1 GPU)
        tops = 9.0 * alg * n / (ms * 1e-3) / 1e12                                                     # 3 x 3 digit products per multiply
        hbm.append({"kernel": "ssg_sqdist_self_i8", "bound": "mfma", "what": "exact integer Gram on v_mfma_i32_32x32x32_i8: 3 balanced radix-256 digits per "
                    "feature, 9 digit products per multiply; achieved = executed int8 ops, algorithmic_tflops = the 2*d flop per distance it replaces "
                    "(fp64 MFMA peak for that: 78.6)", "launches": n, "avg_launch_ms": round(ms / n, 4), "achieved": round(tops, 1),
                    "peak": PEAK_INT8_MFMA_TOPS, "unit": "TOP/s", "frac": round(tops / PEAK_INT8_MFMA_TOPS, 4),
                    "algorithmic_tflops": round(alg * n / (ms * 1e-3) / 1e12, 1)})
    if "ssg_source_rowmin_filtered" in tot:
        n, ms = tot["ssg_source_rowmin_filtered"]
        tf = 2.0 * nrows * Ns * d * n / (ms * 1e-3) / 1e12
        hbm.append({"kernel": "ssg_source_rowmin_filtered", "bound": "mfma", "what": "source term by filter-and-refine: split-half fp16-MFMA bound pass (2*N*Ns*d flop, 3 products each) + fp64 "
                    "re-evaluation of candidate granules; time covers both; peak = fp16 MFMA / 3", "launches": n, "avg_launch_ms": round(ms / n, 4), "achieved": round(tf, 2),
                    "peak": round(PEAK_FP16_MFMA_TF / 3.0, 1), "unit": "TFLOP/s", "frac": round(tf / (PEAK_FP16_MFMA_TF / 3.0), 4)})
    if "ssg_source_rowmin_filtered1" in tot:
        n, ms = tot["ssg_source_rowmin_filtered1"]
        tf = 2.0 * nrows * Ns * d * n / (ms * 1e-3) / 1e12
        hbm.append({"kernel": "ssg_source_rowmin_filtered1", "bound": "mfma", "what": "source term by filter-and-refine: bound pass = plain fp16 GEMM on half copies of the "
                    "operands (2*N*Ns*d flop, one v_mfma_f32_32x32x16_f16 product per term) + fp64 re-evaluation of the candidate granules; time covers the encode, "
                    "both passes and the row norms; peak = dense fp16 MFMA", "launches": n, "avg_launch_ms": round(ms / n, 4), "achieved": round(tf, 2),
                    "peak": PEAK_FP16_MFMA_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP16_MFMA_TF, 4)})
    hbm_ms = sum(tot[k][1] for k in K5_K12 if k in tot) / steps
    k5_12 = 8.0 * nrows * N / (hbm_ms * 1e-3) / 1e9 if hbm_ms > 0 else float("nan")
    k5 = {"bound": "hbm", "achieved": round(k5_12, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(k5_12 / PEAK_HBM_GBS, 4),
          "algorithmic": "8*N^2 bytes per split over all K5..K12 kernel time (SURVEY.md 8d); K5 = the introsort replay unless rank_mode is 'stable'",
          "kernel_ms": round(hbm_ms, 3)}
    return hbm, k5
