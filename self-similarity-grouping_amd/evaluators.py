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
import os
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


class TensorBatchLoader(object):
    """Minimal extraction loader over an image tensor that is already resident (CPU or HBM): yields
    `(imgs [b,3,H,W], fnames, pids, camids)` in order, like the `DataLoader(Preprocessor(...), shuffle=False)` of
    selftraining.py:49-53.  `shard(rank, world)` returns the loader of this rank's contiguous block of batches."""

    def __init__(self, images, batch_size=128, fnames=None, pids=None, first=0, count=None, base=0):
        # base: index (in the loader's item numbering) of images[0] -- a rank that holds only ITS block of a sharded set passes the block
        # with base = the block's first item and count = the size of the whole set; items outside the resident block cannot be iterated
        self.images, self.batch_size = images, int(batch_size)
        self.first, self.base = int(first), int(base)
        self.count = int(images.shape[0] + self.base - first if count is None else count)
        self.fnames, self.pids = fnames, pids

    def __len__(self):
        return (self.count + self.batch_size - 1) // self.batch_size

    def num_items(self):
        return self.count

    def shard(self, rank, world):
        """this rank's contiguous block of IMAGES (`shard_bounds` over the items, not over the batches: 13 batches over 8 ranks would
        leave three ranks with half the work), re-batched locally -- an image's features do not depend on its batch"""
        from .dist import shard_bounds
        i0, i1 = shard_bounds(self.count, rank, world)
        return TensorBatchLoader(self.images, self.batch_size, self.fnames, self.pids, self.first + i0, i1 - i0, self.base)

    def listing(self):
        """(fnames, pids) of every item in loader order, without touching the images"""
        r = range(self.first, self.first + self.count)
        names = list(self.fnames[self.first:self.first + self.count]) if self.fnames is not None else ["%08d" % i for i in r]
        ids = list(self.pids[self.first:self.first + self.count]) if self.pids is not None else [0] * self.count
        return names, ids

    def __iter__(self):
        if self.count > 0 and (self.first < self.base or self.first + self.count > self.base + self.images.shape[0]):
            raise IndexError("TensorBatchLoader: items [%d, %d) requested, the resident block holds [%d, %d)"
                             % (self.first, self.first + self.count, self.base, self.base + self.images.shape[0]))
        for b0 in range(self.first, self.first + self.count, self.batch_size):
            b1 = min(b0 + self.batch_size, self.first + self.count)
            names = self.fnames[b0:b1] if self.fnames is not None else ["%08d" % i for i in range(b0, b1)]
            ids = self.pids[b0:b1] if self.pids is not None else [0] * (b1 - b0)
            yield self.images[b0 - self.base:b1 - self.base], names, ids, [0] * (b1 - b0)


def _rank_batches(data_loader, group):
    """this rank's share of an extraction loader: the contiguous `shard_bounds` block of its batches (the
    reference splits every batch over the GPUs with nn.DataParallel, selftraining.py:135).  Loaders with a `shard(rank, world)`
    method (GpuBatchLoader, TensorBatchLoader) only ever touch their own images; any other iterable is walked in full and the
    foreign batches are skipped (their CPU-side decode is then wasted: give such loaders a `shard`)."""
    if group is None:
        return data_loader
    import torch.distributed as dist
    from .dist import shard_bounds
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if hasattr(data_loader, "shard"):
        return data_loader.shard(rank, world)
    lo, hi = shard_bounds(len(data_loader), rank, world)

    def mine():
        for i, batch in enumerate(data_loader):
            if i >= hi:
                break
            if i >= lo:
                yield batch
    return mine()


def extract_embeddings(model, data_loader, for_eval=False, print_freq=0, group=None, gather=True):
    """Device-resident extraction: returns (feats, fnames, pids) with feats
    [(S+1), N, 2048] (for_eval=False, split model) or [N, D] CUDA float32, in loader order.

    group: torch.distributed group (one process per GPU).  The loader's batches are sharded contiguously over the ranks, every
    rank embeds its own share and the embeddings are all-gathered (C1 of SURVEY.md 8e: ONE flat all-gather of the feature
    blocks over RCCL/xGMI; block lengths and file names / pids follow from the loader description every rank holds -- loaders
    without `shard` / `listing` exchange them with a small all-gather and one `all_gather_object`), so
    every rank returns the full set in loader order -- bit-identical to the unsharded call, because an image's features do not
    depend on which other images share its launch.  gather=False keeps the local share (feats of this rank's images only)."""
    m = _check_model(model).eval()
    chunks, fnames, pids = [], [], []
    t0 = time.time()
    mine = _rank_batches(data_loader, group)
    nb = len(mine) if hasattr(mine, "__len__") else len(data_loader)
    # the split-half range flag (|activation| >= 65520, resnet.ResNet._overflowed) is sticky on the device: loaders that can be walked
    # again read it ONCE after the last batch instead of once per batch (a host round trip that drains the launch queue every time);
    # one-shot iterables keep the per-batch check
    # The deferred check replays the loader, so it is only used for loaders that declare a fixed order (`listing`: TensorBatchLoader,
    # GpuBatchLoader); a torch DataLoader has __len__ too but may shuffle: per-batch check there.
    again = (hasattr(mine, "__len__") and hasattr(mine, "listing") and hasattr(m, "_overflowed")
             and os.environ.get("SSG_EXTRACT_CHECK_EACH", "0") != "1")
    known = hasattr(data_loader, "shard") and hasattr(data_loader, "num_items") and hasattr(data_loader, "listing")
    # C1 overlapped with the forwards (round 5): with a loader description every rank holds, batch i of every rank is all-gathered as
    # soon as it is embedded -- asynchronously, on the communicator's own stream -- while batch i + 1 runs; the pieces are put into
    # loader order at the end.  (One flat gather after the last batch left the xGMI links idle during the whole extraction.)
    ov = None
    if group is not None and gather and known and hasattr(mine, "batch_size") and os.environ.get("SSG_EXTRACT_OVERLAP", "1") != "0":
        import torch.distributed as dist
        from .dist import _flat_gather_supported
        if _flat_gather_supported(group):
            world = dist.get_world_size(group)
            ov = {"world": world, "bs": int(mine.batch_size), "counts": [int(data_loader.shard(r, world).num_items()) for r in range(world)], "recv": [], "work": []}
            ov["nb"] = max((c + ov["bs"] - 1) // ov["bs"] for c in ov["counts"])

    def image_major(f):
        return f.permute(1, 0, 2).contiguous() if f.dim() == 3 else f

    def post(rows, tail=None):
        """queue the all-gather of one batch's rows (padded to the batch size; `rows` None: this rank has no batch i, zeros travel)"""
        import torch.distributed as dist
        send = torch.zeros((ov["bs"],) + tuple(tail if rows is None else rows.shape[1:]), dtype=torch.float32, device=m.device)
        if rows is not None:
            send[: rows.shape[0]] = rows
        recv = torch.empty((ov["world"] * ov["bs"],) + tuple(send.shape[1:]), dtype=torch.float32, device=m.device)
        ov["work"].append(dist.all_gather_into_tensor(recv, send, group=group, async_op=True))
        ov["recv"].append(recv)

    for i, batch in enumerate(mine):
        imgs, names, ids = batch[0], batch[1], batch[2]
        chunks.append(m.embed_with_flip(torch.as_tensor(imgs), for_eval=for_eval, check_overflow=False) if again
                      else m.embed_with_flip(torch.as_tensor(imgs), for_eval=for_eval))
        fnames.extend(list(names)); pids.extend(list(ids))
        if ov is not None:
            post(image_major(chunks[-1]))
        if print_freq and (i + 1) % print_freq == 0:
            print('Extract Features: [{}/{}]\tTime {:.3f}'.format(i + 1, nb, time.time() - t0))
    nsets = (m.num_split + 1) if m.num_split > 1 else 1
    three = nsets > 1 and not for_eval
    if ov is not None:
        for _ in range(len(chunks), ov["nb"]):                 # ranks with fewer batches still take part in every collective
            post(None, (nsets, 2048) if three else (nsets * 2048,))
    redo = bool(again and m._overflowed()) if ov is None else False
    if ov is not None and again:
        # the range flag of EVERY rank (one small gather + the one read the deferred check costs anyway): the pieces already travelled,
        # so a rank that has to recompute makes every rank gather again
        from .dist import gather_rows
        flag = m._ovf.view(1, 1) if (m.precision == "split" and getattr(m, "_ovf", None) is not None) else torch.zeros((1, 1), dtype=torch.int32, device=m.device)
        flags = [int(x) for x in gather_rows(flag, group).flatten().tolist()]
        import torch.distributed as dist
        redo = bool(flags[dist.get_rank(group)])
        if redo:
            m._ovf.zero_()
        if any(flags):
            for w in ov["work"]:
                w.wait()
            ov = None                                           # fall back to the single gather below, on recomputed features
    if redo:          # (the fp32 twin warns when it is first built)
        chunks.clear()                     # the first pass's features are discarded BEFORE the second pass allocates its own
        chunks = [m._f32_twin().embed_with_flip(torch.as_tensor(batch[0]), for_eval=for_eval) for batch in mine]
    if chunks:
        feats = torch.cat(chunks, dim=1 if chunks[0].dim() == 3 else 0)
    else:       # more ranks than batches: an empty share of the right shape
        feats = torch.empty((nsets, 0, 2048) if three else (0, nsets * 2048), dtype=torch.float32, device=m.device)
    if group is None or not gather:
        return feats, fnames, pids
    import torch.distributed as dist
    from .dist import gather_counts, gather_ragged
    world = dist.get_world_size(group)
    if ov is not None:
        for w in ov["work"]:
            w.wait()
        bs, parts = ov["bs"], []
        for r, c in enumerate(ov["counts"]):                    # loader order = rank order, batch order inside a rank
            for i in range((c + bs - 1) // bs):
                parts.append(ov["recv"][i][r * bs: r * bs + min(bs, c - i * bs)])
        rows = torch.cat(parts, dim=0) if parts else image_major(feats)
        feats = rows.permute(1, 0, 2).contiguous() if three else rows
        fnames, pids = data_loader.listing()
        return feats, fnames, pids
    rows = image_major(feats)          # image-major rows
    if known:
        # every rank holds the same loader description: block lengths and names follow from it, no exchange and no host round trip
        counts = [int(data_loader.shard(r, world).num_items()) for r in range(world)]
        if counts[dist.get_rank(group)] != rows.shape[0]:
            raise RuntimeError("extract_embeddings: this rank embedded %d images, its shard of the loader has %d" % (rows.shape[0], counts[dist.get_rank(group)]))
    else:
        counts = gather_counts(rows.shape[0], group, rows.device)
    rows = gather_ragged(rows, counts, group)
    feats = rows.permute(1, 0, 2).contiguous() if three else rows
    if known:
        fnames, pids = data_loader.listing()
    else:
        meta = [None] * world
        dist.all_gather_object(meta, (fnames, pids), group=group)
        fnames = [f for part in meta for f in part[0]]
        pids = [p for part in meta for p in part[1]]
    return feats, fnames, pids


def extract_features(model, data_loader, print_freq=20, for_eval=True, metric=None, group=None):
    """Drop-in for reid/evaluators.py:18-60.  group: see extract_embeddings (sharded over the GPUs of the node, full
    dictionaries on every rank)."""
    m = _check_model(model)
    feats, fnames, pids = extract_embeddings(m, data_loader, for_eval=for_eval, print_freq=print_freq, group=group)
    # the finiteness test runs on the device (one scalar comes back), the features travel once into page-locked memory (hostio.py) and the
    # per-image views are made by unbind (one C++ loop) -- the dictionaries cost ~3 % on top of the device-resident extraction, not 10 %
    from . import hostio
    finite = torch.isfinite(feats).all() if feats.is_cuda else None
    cpu = hostio.to_host(feats) if feats.is_cuda else feats
    if not bool(finite if finite is not None else torch.isfinite(cpu).all()):
        # (the split-half path detects half-range overflow itself and recomputes such batches in fp32: resnet.ResNet._overflowed)
        from ._lib import SSGError
        raise SSGError("non-finite embeddings (NaN/inf in the input images or the weights?)")
    if cpu.dim() == 3:       # split model, for_eval=False: list of S+1 vectors per image (evaluators.py:37-39)
        per_set = [cpu[s].unbind(0) for s in range(cpu.shape[0])]
        features = OrderedDict(zip(fnames, (list(t) for t in zip(*per_set))))
    else:
        features = OrderedDict(zip(fnames, cpu.unbind(0)))
    labels = OrderedDict(zip(fnames, pids))
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
