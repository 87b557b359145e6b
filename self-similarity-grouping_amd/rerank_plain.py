"""kNN-set Jaccard re-ranking -- host-side mirror of reid/rerank_plain.py:125-178 `re_ranking`.

`re_ranking(input_feature_source, input_feature, k=20, lambda_value=0.1, MemorySave=False, Minibatch=2000)` keeps the
reference signature and returns `(final_dist, final_dist)` as float64 numpy arrays (the reference returns the same
array twice).  `re_ranking_plain_device` is the fused path: it returns the same `DistHandle` as
`rerank.re_ranking_device` (half J' + source vector), so `eps_rule` / `DBSCAN` / `generate_selflabel` consume it without
materialising the float64 matrix.

Stages: source term and half original distance exactly as rerank.py (the reference's lines are identical) on the
kernels of that path; the k-th smallest entry of every row from the top-k kernel; set members, inverted index and
set-Jaccard rows in csrc/rerank_plain.hip.  Single GPU (the variant is not on the sharded benchmark path).
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .rerank import DeviceBackedArray, DistHandle, ReRankNaNError, _as_dev_f32, _original_distance, range_stats, source_vector


def re_ranking_plain_device(src, tgt, k=20, lambda_value=0.1, stages=None, memory_save=False):
    L = _lib.lib()
    dev = tgt.device if torch.is_tensor(tgt) and tgt.is_cuda else torch.device("cuda", torch.cuda.current_device())
    src = _as_dev_f32(src, dev); tgt = _as_dev_f32(tgt, dev)
    N, d = tgt.shape
    if not (1 <= k <= min(N, 64)):
        raise ValueError("re_ranking (plain): need 1 <= k <= min(N, 64), got k=%d N=%d" % (k, N))
    st = stream()
    stats = range_stats(tgt, src)
    D, rowmax, flag = _original_distance(L, tgt, 0, N, stats[0], st, memory_save)
    # source-domain term (rerank_plain.py:130-143)
    rowmin = source_vector(src, tgt, 0, N, stats=stats)
    v = torch.empty(N, dtype=torch.float16, device=dev)
    vmax = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.ssg_source_vec_finish(ptr(rowmin), N, ptr(v), ptr(vmax), st), "ssg_source_vec_finish")
    # k-th smallest entry of every row (np.partition(tem_vec, k-1)[k-1], :167): the top-k list ordered by raw value
    ones = torch.full((N,), 0x3C00, dtype=torch.int32, device=dev)      # rowmax = half(1): keys are the raw values
    rank = torch.empty((N, k), dtype=torch.int32, device=dev)
    check(L.ssg_topk_rank(ptr(D), ptr(ones), N, N, k, ptr(rank), st), "ssg_topk_rank")
    cap = max(64, 4 * k)
    while True:
        a_idx = torch.empty((N, cap), dtype=torch.int32, device=dev); a_val = torch.empty((N, cap), dtype=torch.float16, device=dev)
        a_nnz = torch.empty(N, dtype=torch.int32, device=dev); ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        check(L.ssg_knn_sets(ptr(D), ptr(rank), N, 0, N, k, cap, ptr(a_idx), ptr(a_val), ptr(a_nnz), ptr(ovf), st), "ssg_knn_sets")
        total, over = torch.stack([a_nnz.sum(), ovf[0].to(torch.int64)]).tolist()
        if not over:
            break
        cap = min(N, cap * 8)          # many exact ties at the k-th distance: retry with room for them
    total = int(total)
    colcnt = torch.empty(N, dtype=torch.int32, device=dev); colptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
    inv_row = torch.empty(max(total, 1), dtype=torch.int32, device=dev); inv_val = torch.empty(max(total, 1), dtype=torch.float16, device=dev)
    check(L.ssg_invert_index(ptr(a_idx), ptr(a_val), ptr(a_nnz), N, N, cap, ptr(colcnt), ptr(colptr), ptr(inv_row), ptr(inv_val), st),
          "ssg_invert_index")
    om = L.ssg_double_to_half_bits(1.0 - float(lambda_value))
    Jp = torch.empty((N, N), dtype=torch.float16, device=dev)
    check(L.ssg_set_jaccard_rows(ptr(a_idx), ptr(a_nnz), cap, ptr(colptr), ptr(inv_row), N, 0, N, om, ptr(Jp), st), "ssg_set_jaccard_rows")
    vmax_h, flag_h = torch.cat([vmax, flag if flag is not None else torch.zeros_like(vmax)]).tolist()
    if flag_h:
        raise _lib.SSGError("ssg_gram_i8_encode: a feature did not fit the digit count chosen from max|feat| (internal error)")
    if int(vmax_h) & 0x7FFF == 0:
        raise ReRankNaNError("max(source_dist_vec) == 0: the reference (reid/rerank_plain.py:135) would return an all-NaN final_dist")
    if stages is not None:
        stages.update(D=D, v=v, rank=rank, a_idx=a_idx, a_nnz=a_nnz, Jp=Jp)
    return DistHandle(N, 0, Jp, v=v, lambda_value=lambda_value, euclid=D)


def re_ranking(input_feature_source, input_feature, k=20, lambda_value=0.1, MemorySave=False, Minibatch=2000, device=None):
    """Drop-in for reid/rerank_plain.py:125 re_ranking (numpy in, numpy out; both returns are the final distance)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    print('computing source distance...')
    print('computing original distance...')
    h = re_ranking_plain_device(_as_dev_f32(np.asarray(input_feature_source), device), _as_dev_f32(np.asarray(input_feature), device),
                                k=k, lambda_value=lambda_value, memory_save=MemorySave)
    final = DeviceBackedArray.attach(h.final_dist().cpu().numpy(), h)
    return final, final
