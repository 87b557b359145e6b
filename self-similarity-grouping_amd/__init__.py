"""ssg_amd -- MI355X-native pseudo-label grouping for Self-Similarity Grouping.

One hot path: ResNet-50 embedding of the target set -> N x N
k-reciprocal re-rank distance -> epsilon rule -> DBSCAN, behind the reference's Python call
surface.  Compute = hand-written HIP kernels in csrc/ behind the C ABI of include/ssg_hip.h;
this package is the thin host-side mirror of the reference interface.
"""
from . import _lib  # noqa: F401
from ._lib import SSGError, available  # noqa: F401

__all__ = ["re_ranking", "re_ranking_device", "DBSCAN", "eps_rule", "eps_rule_dbscan", "compute_dist", "generate_selflabel", "SSGError", "available"]


def __getattr__(name):   # lazy: torch import only when the compute surface is touched
    if name in ("re_ranking", "re_ranking_device", "re_ranking_init", "re_ranking_init_dist", "DistHandle", "ReRankNaNError"):
        from . import rerank
        return getattr(rerank, name)
    if name in ("DBSCAN", "eps_rule", "eps_rule_dbscan", "as_handle"):
        from . import cluster
        return getattr(cluster, name)
    if name in ("compute_dist", "generate_selflabel", "select_labeled", "generate_dataset"):
        from . import selftraining
        return getattr(selftraining, name)
    if name in ("extract_features", "extract_embeddings", "extract_cnn_feature", "fliplr", "pairwise_distance", "pairwise_distance_device",
                "TensorBatchLoader"):
        from . import evaluators
        return getattr(evaluators, name)
    if name in ("re_ranking_plain", "re_ranking_plain_device"):
        from . import rerank_plain
        return rerank_plain.re_ranking if name == "re_ranking_plain" else rerank_plain.re_ranking_plain_device
    if name in ("cmc", "mean_ap", "evaluate_all", "Evaluator"):
        from . import ranking
        return getattr(ranking, name)
    if name in ("Preprocessor", "GpuBatchLoader", "preprocess_batch"):
        from . import preprocessor
        return getattr(preprocessor, name)
    if name == "decode_jpeg_batch":
        from . import jpeg
        return jpeg.decode_batch
    if name == "triplet_pairwise_dist":
        from . import triplet
        return triplet.pairwise_dist
    if name in ("create", "ResNet", "synthetic_state_dict"):
        from . import resnet
        return getattr(resnet, name)
    raise AttributeError(name)
