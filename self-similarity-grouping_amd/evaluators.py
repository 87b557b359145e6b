"""Feature-extraction harness -- host-side mirror of reid/evaluators.py:12-85 and
reid/feature_extraction/cnn.py:10-22.

`extract_features(model, data_loader, print_freq=20, for_eval=True, metric=None)` keeps the
reference signature and returns `(OrderedDict fname -> features, OrderedDict fname -> pid)`
with CPU tensors exactly like the reference (a list of S+1 unit-norm [2048] vectors per image
for split models with for_eval=False, one vector otherwise).  Both orientations of a batch
(original + horizontally flipped, evaluators.py:28-35) run as one fused device pass; features
stay on the GPU until the dictionary is built, so there is one D2H copy per call instead of
one per batch.  `extract_embeddings` is the device-resident variant used by the fused
grouping path (no dict, no D2H).
"""
import time
from collections import OrderedDict

import torch

from .resnet import ResNet


def fliplr(img):
    """flip horizontal (evaluators.py:12-16); kept for API compatibility -- the extractor
    fuses the flip into its NCHW->NHWC input kernel instead of materialising it."""
    inv_idx = torch.arange(img.size(3) - 1, -1, -1, device=img.device).long()
    return img.index_select(3, inv_idx)


def extract_cnn_feature(model, inputs, for_eval, modules=None):
    """reid/feature_extraction/cnn.py:10-22: model.eval(); model(inputs, for_eval)[0] -> CPU."""
    if modules is not None:
        raise NotImplementedError("forward hooks (cnn.py:24-35) are not part of the grouping path")
    model.eval()
    with torch.no_grad():
        outputs = model(torch.as_tensor(inputs), for_eval)[0]
    if isinstance(outputs, list):
        return [x.data.cpu() for x in outputs]
    return outputs.data.cpu()


def _check_model(model):
    m = getattr(model, "module", model)
    if not isinstance(m, ResNet):
        raise TypeError("ssg_amd.extract_features drives ssg_amd.resnet.ResNet embedders (HIP kernels); got %r" % type(model))
    return m


def extract_embeddings(model, data_loader, for_eval=False, print_freq=0):
    """Device-resident extraction: returns (feats, fnames, pids) with feats
    [(S+1), N, 2048] (for_eval=False, split model) or [N, D] CUDA float32, in loader order."""
    m = _check_model(model).eval()
    chunks, fnames, pids = [], [], []
    t0 = time.time()
    for i, batch in enumerate(data_loader):
        imgs, names, ids = batch[0], batch[1], batch[2]
        chunks.append(m.embed_with_flip(torch.as_tensor(imgs), for_eval=for_eval))
        fnames.extend(list(names)); pids.extend(list(ids))
        if print_freq and (i + 1) % print_freq == 0:
            print('Extract Features: [{}/{}]\tTime {:.3f}'.format(i + 1, len(data_loader), time.time() - t0))
    feats = torch.cat(chunks, dim=1 if chunks[0].dim() == 3 else 0)
    return feats, fnames, pids


def extract_features(model, data_loader, print_freq=20, for_eval=True, metric=None):
    """Drop-in for reid/evaluators.py:18-60."""
    m = _check_model(model)
    feats, fnames, pids = extract_embeddings(m, data_loader, for_eval=for_eval, print_freq=print_freq)
    features, labels = OrderedDict(), OrderedDict()
    cpu = feats.cpu()
    if not bool(torch.isfinite(cpu).all()):
        # (the split-half path detects half-range overflow itself and recomputes such batches in fp32: resnet.ResNet._overflowed)
        from ._lib import SSGError
        raise SSGError("non-finite embeddings (NaN/inf in the input images or the weights?)")
    if cpu.dim() == 3:       # split model, for_eval=False: list of S+1 vectors per image (evaluators.py:37-39)
        for idx, (fname, pid) in enumerate(zip(fnames, pids)):
            features[fname] = [cpu[s, idx] for s in range(cpu.shape[0])]
            labels[fname] = pid
    else:
        for idx, (fname, pid) in enumerate(zip(fnames, pids)):
            features[fname] = cpu[idx]
            labels[fname] = pid
    return features, labels


def _sqdist(x, y, self_form=False):
    """float32 squared-L2 block on the HIP fp32-MFMA GEMM (ssg_pairwise_sqdist_f32)."""
    from . import _lib
    from ._lib import check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = x.to(dev, torch.float32).contiguous(); y = y.to(dev, torch.float32).contiguous()
    m, d = x.shape; n = y.shape[0]
    dp, npad = (-d) % 32, (-n) % 64
    if dp:
        x = torch.nn.functional.pad(x, (0, dp)); y = torch.nn.functional.pad(y, (0, dp))
    if npad:
        y = torch.nn.functional.pad(y, (0, 0, 0, npad))
    out = torch.empty((m, n + npad), dtype=torch.float32, device=dev)
    ws = torch.empty(m + n + npad, dtype=torch.float32, device=dev)
    check(L.ssg_pairwise_sqdist_f32(ptr(x), ptr(y), m, n + npad, d + dp, 1 if self_form else 0, ptr(ws), ptr(out), stream()), "ssg_pairwise_sqdist_f32")
    return out[:, :n]


def pairwise_distance(features, query=None, gallery=None, metric=None):
    """reid/evaluators.py:63-85 on the GPU (float32 squared L2; returns a CPU tensor like the
    reference).  The query=None branch keeps the reference's 2|x_i|^2 - 2<x_i,x_j> form (:64-72),
    which equals the squared distance only for unit-norm rows."""
    return pairwise_distance_device(features, query, gallery, metric).cpu()


def pairwise_distance_device(features, query=None, gallery=None, metric=None):
    """pairwise_distance with the result left on the GPU (consumed by ssg_amd.ranking.evaluate_all)."""
    if query is None and gallery is None:
        n = len(features)
        x = torch.cat([f.view(1, -1) for f in features.values()]).view(n, -1)
        if metric is not None:
            x = metric.transform(x)
        return _sqdist(x, x, self_form=True)
    x = torch.cat([features[f].unsqueeze(0) for f, _, _ in query], 0)
    y = torch.cat([features[f].unsqueeze(0) for f, _, _ in gallery], 0)
    m, n = x.size(0), y.size(0)
    x = x.view(m, -1); y = y.view(n, -1)
    if metric is not None:
        x = metric.transform(x); y = metric.transform(y)
    return _sqdist(x, y)
