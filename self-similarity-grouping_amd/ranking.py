"""Retrieval metrics -- host-side mirror of reid/evaluation_metrics/ranking.py:18-115 (cmc, mean_ap) and
reid/evaluators.py:88-129 (evaluate_all), :147-166 (Evaluator), computed on the GPU (csrc/ranking.hip).

`cmc(...)`, `mean_ap(...)` and `evaluate_all(...)` keep the reference signatures and return types (numpy
cumulative-match curve, float mAP, CMC top-1).  The distance block may be a numpy array, a CPU tensor or a CUDA
tensor (the Evaluator keeps it on the device).  Supported protocols: the deterministic ones -- 'market1501'
(`first_match_break=True`, what evaluate_all uses) and 'allshots' (every match credited with 1/len), each optionally with
`separate_camera_set`; the `single_gallery_shot` protocol draws random gallery subsets from numpy's global RNG
(ranking.py:11-16,56-61; commented out in the reference's evaluate_all) and raises NotImplementedError.

Equal distances are ordered by gallery index (numpy's default argsort leaves the tie order unspecified); mAP
does not depend on it, and CMC only when a true match ties with a non-match.  No CPU fallback.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def _dev():
    if not torch.cuda.is_available():
        raise _lib.SSGError("ssg_amd.ranking runs on the GPU only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _ids(x, n, default, dev):
    if x is None:
        x = default(n)
    return torch.as_tensor(np.asarray(x).astype(np.int64)).to(torch.int32).to(dev).contiguous()


def per_query(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None, separate_camera_set=False):
    """-> (first_rank int32[m] CUDA, ap float64[m] CUDA); -1 / NaN mark queries without a valid true match."""
    L = _lib.lib()
    dev = _dev()
    d = torch.as_tensor(distmat).to(dev, torch.float32)
    if d.dim() != 2:
        raise ValueError("distmat must be [m, n]")
    if d.stride(1) != 1:
        d = d.contiguous()
    m, n = d.shape
    qid = _ids(query_ids, m, np.arange, dev); gid = _ids(gallery_ids, n, np.arange, dev)
    qcam = _ids(query_cams, m, lambda k: np.zeros(k), dev); gcam = _ids(gallery_cams, n, lambda k: np.ones(k), dev)
    if qid.numel() != m or qcam.numel() != m or gid.numel() != n or gcam.numel() != n:
        raise ValueError("id / camera lists do not match the distance block")
    first = torch.empty(m, dtype=torch.int32, device=dev); ap = torch.empty(m, dtype=torch.float64, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.ssg_rank_metrics(ptr(d), m, n, d.stride(0), ptr(qid), ptr(qcam), ptr(gid), ptr(gcam), 1 if separate_camera_set else 0,
                             ptr(first), ptr(ap), ptr(ovf), stream()), "ssg_rank_metrics")
    if int(ovf.item()):
        raise _lib.SSGError("ssg_rank_metrics: a query has more than 2048 true matches in the gallery")
    return first, ap


def per_query_all(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None, separate_camera_set=False):
    """-> (nmatch int32[m], nm_before int32[m, cap]) numpy: per query the number of valid true matches and, for its s-th match,
    the number of valid non-matching gallery entries ordered before it (the bins the all-shots CMC credits)."""
    L = _lib.lib()
    dev = _dev()
    d = torch.as_tensor(distmat).to(dev, torch.float32)
    if d.dim() != 2:
        raise ValueError("distmat must be [m, n]")
    if d.stride(1) != 1:
        d = d.contiguous()
    m, n = d.shape
    qid = _ids(query_ids, m, np.arange, dev); gid = _ids(gallery_ids, n, np.arange, dev)
    qcam = _ids(query_cams, m, lambda k: np.zeros(k), dev); gcam = _ids(gallery_cams, n, lambda k: np.ones(k), dev)
    cap = int(min(2048, max(1, int(torch.unique(gid, return_counts=True)[1].max().item())))) if n else 1      # no query can have more matches than the largest
    # gallery identity (unique counts, not bincount: raw Market-1501 lists carry the junk pid -1, ids may be large and sparse)
    first = torch.empty(m, dtype=torch.int32, device=dev); ap = torch.empty(m, dtype=torch.float64, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    nmb = torch.zeros((m, cap), dtype=torch.int32, device=dev); nm = torch.zeros(m, dtype=torch.int32, device=dev)
    check(L.ssg_rank_metrics_all(ptr(d), m, n, d.stride(0), ptr(qid), ptr(qcam), ptr(gid), ptr(gcam), 1 if separate_camera_set else 0,
                                 ptr(first), ptr(ap), ptr(ovf), ptr(nmb), ptr(nm), cap, stream()), "ssg_rank_metrics_all")
    if int(ovf.item()):
        raise _lib.SSGError("ssg_rank_metrics: a query has more than 2048 true matches in the gallery")
    return nm.cpu().numpy(), nmb.cpu().numpy()


def cmc(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None, topk=100,
        separate_camera_set=False, single_gallery_shot=False, first_match_break=False):
    """ranking.py:18-79 for the deterministic protocols."""
    if single_gallery_shot:
        raise NotImplementedError("single_gallery_shot draws random gallery subsets from numpy's global RNG (ranking.py:11-16,56-61); "
                                  "the deterministic protocols (first_match_break True/False, separate_camera_set) run on the GPU path")
    if not first_match_break:
        nm, nmb = per_query_all(distmat, query_ids, gallery_ids, query_cams, gallery_cams, separate_camera_set)
        valid = nm > 0
        if not valid.any():
            raise RuntimeError("No valid query")
        ret = np.zeros(topk)
        for q in np.nonzero(valid)[0]:           # sequential adds in query / rank order, like the reference loop (ranking.py:68-75)
            bins = nmb[q, :nm[q]]
            bins = bins[: int(np.searchsorted(bins, topk, side="left"))]
            np.add.at(ret, bins, 1. / nm[q])
        return ret.cumsum() / int(valid.sum())
    first, _ = per_query(distmat, query_ids, gallery_ids, query_cams, gallery_cams, separate_camera_set)
    first = first.cpu().numpy()
    valid = first >= 0
    if not valid.any():
        raise RuntimeError("No valid query")
    ret = np.zeros(topk)
    hit = first[valid & (first < topk)]
    np.add.at(ret, hit, 1)
    return ret.cumsum() / int(valid.sum())


def mean_ap(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None):
    """ranking.py:82-115."""
    _, ap = per_query(distmat, query_ids, gallery_ids, query_cams, gallery_cams)
    ap = ap.cpu().numpy()
    aps = ap[~np.isnan(ap)]
    if aps.size == 0:
        raise RuntimeError("No valid query")
    return np.mean(aps)


def evaluate_all(distmat, query=None, gallery=None, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None,
                 cmc_topk=(1, 5, 10)):
    """reid/evaluators.py:88-129: prints mAP and the market1501 CMC scores, returns CMC top-1."""
    if query is not None and gallery is not None:
        query_ids = [pid for _, pid, _ in query]; gallery_ids = [pid for _, pid, _ in gallery]
        query_cams = [cam for _, _, cam in query]; gallery_cams = [cam for _, _, cam in gallery]
    else:
        assert (query_ids is not None and gallery_ids is not None and query_cams is not None and gallery_cams is not None)
    first, ap = per_query(distmat, query_ids, gallery_ids, query_cams, gallery_cams)     # one device pass serves both metrics
    first = first.cpu().numpy(); ap = ap.cpu().numpy()
    valid = first >= 0
    if not valid.any():
        raise RuntimeError("No valid query")
    mAP = np.mean(ap[valid])
    print('Mean AP: {:4.1%}'.format(mAP))
    ret = np.zeros(100)
    np.add.at(ret, first[valid & (first < 100)], 1)
    scores = ret.cumsum() / int(valid.sum())
    print('CMC Scores{:>12}'.format('market1501'))
    for k in cmc_topk:
        print('top-{:<4}{:12.1%}'.format(k, scores[k - 1]))
    return scores[0]


class Evaluator(object):
    """reid/evaluators.py:147-166: features -> query x gallery distance block -> metrics, all on the GPU."""

    def __init__(self, model, print_freq=1):
        self.model = model
        self.print_freq = print_freq

    def evaluate(self, data_loader, query, gallery, metric=None):
        from .evaluators import extract_features, pairwise_distance_device
        features, _ = extract_features(self.model, data_loader, print_freq=self.print_freq)
        distmat = pairwise_distance_device(features, query, gallery, metric=metric)
        return evaluate_all(distmat, query=query, gallery=gallery)
