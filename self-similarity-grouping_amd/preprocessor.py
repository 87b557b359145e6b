"""Input side of the extraction loaders on the GPU -- host-side mirror of reid/utils/data/preprocessor.py:7-30 with the
transform of selftraining.py:43-47 / :66-70 (`Resize((height, width))`, `ToTensor()`, `Normalize(mean, std)`).

The reference decodes AND transforms every image on the CPU inside DataLoader workers.  Here the JPEG decode stays on the
CPU (PIL, like the reference: `Image.open(fpath).convert('RGB')`), the resize + tensor conversion + normalisation run as
two streaming HIP kernels on a batch of decoded images (`ssg_preprocess_u8`), bit-exact with Pillow's own resize.

`Preprocessor` keeps the reference's constructor and item layout `(img, fname, pid, camid)`; `GpuBatchLoader` is the
replacement for the `DataLoader(Preprocessor(...), batch_size, shuffle=False)` pair that `extract_features` iterates: it
yields `(imgs [B,3,H,W] float32 CUDA, fnames, pids, camids)` in dataset order.
"""
import math
import os.path as osp

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

MEAN = (0.485, 0.456, 0.406)      # selftraining.py:38-39
STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2      # Pillow libImaging/Resample.c
_coeff_cache = {}


def bilinear_coeffs(in_size, out_size):
    """Pillow's resampling windows for the triangle filter (Resample.c precompute_coeffs + normalize_coeffs_8bpc):
    -> (first input index [out], window length [out], 22-bit fixed-point coefficients [out, ksize]) as int32 arrays."""
    key = (int(in_size), int(out_size))
    if key in _coeff_cache:
        return _coeff_cache[key]
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = filterscale                      # bilinear: support 1.0, widened by the down-scale factor
    ksize = int(math.ceil(support)) * 2 + 1
    first = np.zeros(out_size, np.int32); count = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = []
        total = 0.0
        for x in range(lo, hi):
            a = abs((x - center + 0.5) * inv)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v); total += v
        for t, v in enumerate(w):
            if total != 0.0:
                v = v / total
            kk[xx, t] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        first[xx] = lo; count[xx] = hi - lo
    _coeff_cache[key] = (first, count, kk)
    return _coeff_cache[key]


def preprocess_batch(images_u8, height, width, mean=MEAN, std=STD, device=None):
    """uint8 [B, h, w, 3] (numpy / torch, CPU or CUDA; decoded RGB images of one size) -> float32 CUDA [B, 3, height, width]
    == torch.stack([Normalize(mean, std)(ToTensor()(Resize((height, width))(img))) for img in images])."""
    import ctypes
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    x = torch.as_tensor(images_u8)
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("expected uint8 images [B, h, w, 3] (RGB), got %r %r" % (x.dtype, tuple(x.shape)))
    x = x.to(device).contiguous()
    B, h, w, _ = x.shape
    dev_tabs = []
    for n_in, n_out in ((w, width), (h, height)):
        first, count, kk = bilinear_coeffs(n_in, n_out)
        dev_tabs.append((torch.from_numpy(first).to(device), torch.from_numpy(count).to(device), torch.from_numpy(kk).to(device).contiguous(), kk.shape[1]))
    tmp = torch.empty((B, h, width, 3), dtype=torch.uint8, device=device)
    out = torch.empty((B, 3, height, width), dtype=torch.float32, device=device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean]); s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    (xf, xc, xk, xks), (yf, yc, yk, yks) = dev_tabs
    check(L.ssg_preprocess_u8(ptr(x), B, h, w, height, width, ptr(xf), ptr(xc), ptr(xk), xks, ptr(yf), ptr(yc), ptr(yk), yks, m3, s3, ptr(tmp), ptr(out),
                              stream()), "ssg_preprocess_u8")
    return out


class Preprocessor(object):
    """reid/utils/data/preprocessor.py:7-30 with the decode only: items are `(uint8 HWC RGB array, fname, pid, camid)`; the
    transform the reference applies per item runs batched on the GPU (`preprocess_batch` / `GpuBatchLoader`)."""

    def __init__(self, dataset, root=None, transform=None, raw=False):
        if transform is not None:
            raise ValueError("the transform (Resize + ToTensor + Normalize, selftraining.py:43-47) runs on the GPU: pass height/width to GpuBatchLoader")
        self.dataset, self.root, self.transform = dataset, root, None
        self.raw = raw          # True: items carry the file's bytes (decoded on the GPU by GpuBatchLoader) instead of Pillow's decoded array

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, indices):
        if isinstance(indices, (tuple, list)):
            return [self._get_single_item(index) for index in indices]
        return self._get_single_item(indices)

    def _get_single_item(self, index):
        from PIL import Image
        fname, pid, camid = self.dataset[index]
        fpath = fname if self.root is None else osp.join(self.root, fname)
        if self.raw:
            with open(fpath, "rb") as f:
                return f.read(), fname, pid, camid
        img = np.asarray(Image.open(fpath).convert('RGB'))
        return img, fname, pid, camid


class GpuBatchLoader(object):
    """Iterable replacement for `DataLoader(Preprocessor(dataset, root, transform), batch_size, shuffle=False)` in
    selftraining.py:49-53: yields `(imgs [B,3,H,W] float32 CUDA, fnames, pids, camids)` in dataset order.  Images of a
    batch that share a size go through one kernel launch pair (Market-1501: all 128x64)."""

    def __init__(self, dataset, root=None, height=256, width=128, batch_size=128, mean=MEAN, std=STD, device=None, decode="gpu"):
        # decode='gpu': the files' bytes go to the device and are decoded there (ssg_amd.jpeg: baseline JPEG bit-exact with Pillow; anything
        # else falls back to Pillow per file); decode='pillow': every file is decoded on the host like the reference does
        if decode not in ("gpu", "pillow"):
            raise ValueError("decode must be 'gpu' or 'pillow'")
        self.decode = decode
        self.items = dataset if isinstance(dataset, Preprocessor) else Preprocessor(dataset, root, raw=(decode == "gpu"))
        if isinstance(dataset, Preprocessor):
            self.decode = "gpu" if dataset.raw else "pillow"
        self.height, self.width, self.batch_size, self.mean, self.std, self.device = height, width, batch_size, mean, std, device

        self.first, self.count = 0, len(self.items)

    def __len__(self):
        return (self.count + self.batch_size - 1) // self.batch_size

    def shard(self, rank, world):
        """loader of this rank's contiguous block of images (`ssg_amd.dist.shard_bounds` over the items): what
        `extract_features(..., group=)` iterates, so that a rank only decodes the images it embeds"""
        import copy
        from .dist import shard_bounds
        i0, i1 = shard_bounds(self.count, rank, world)      # by images, not by batches: balanced to one image, re-batched locally
        sub = copy.copy(self)
        sub.first, sub.count = self.first + i0, i1 - i0
        return sub

    def num_items(self):
        return self.count

    def listing(self):
        """(fnames, pids) of every item in loader order, from the dataset records alone (no file is opened)"""
        recs = [self.items.dataset[i] for i in range(self.first, self.first + self.count)]
        return [r[0] for r in recs], [r[1] for r in recs]

    def _finish(self, recs, pix, dev):
        """decoded pixels of one batch -> the loader's item (transform on the device)"""
        names = [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs]
        if torch.is_tensor(pix):                               # a batch of equally sized baseline files: straight into the transform
            return (preprocess_batch(pix, self.height, self.width, self.mean, self.std, dev),) + names
        out = torch.empty((len(recs), 3, self.height, self.width), dtype=torch.float32, device=dev)
        by_size = {}
        for j, a in enumerate(pix):
            by_size.setdefault(tuple(a.shape[:2]), []).append(j)
        for _, js in by_size.items():
            batch = torch.stack([pix[j] for j in js]) if self.decode == "gpu" else np.stack([pix[j] for j in js])
            res = preprocess_batch(batch, self.height, self.width, self.mean, self.std, dev)
            out[torch.as_tensor(js, device=dev)] = res
        return (out,) + names

    def __iter__(self):
        n = self.first + self.count
        dev = torch.device("cuda", torch.cuda.current_device()) if self.device is None else torch.device(self.device)
        ahead = None        # GPU decode: (records, PendingDecode) of the batch whose decode is queued but whose status words are unread
        for b0 in range(self.first, n, self.batch_size):
            recs = [self.items[i] for i in range(b0, min(n, b0 + self.batch_size))]
            if self.decode == "gpu":
                # one batch of lookahead: batch k + 1 is parsed on the host and its decode queued BEFORE batch k's status words are
                # looked at, so that look never waits for the GPU (uint8 CUDA [H, W, 3] per file, or ONE [B, H, W, 3] tensor)
                from .jpeg import decode_batch_async
                nxt = (recs, decode_batch_async([r[0] for r in recs], dev))
                if ahead is not None:
                    yield self._finish(ahead[0], ahead[1].result(packed=True), dev)
                ahead = nxt
            else:
                yield self._finish(recs, [r[0] for r in recs], dev)
        if ahead is not None:
            yield self._finish(ahead[0], ahead[1].result(packed=True), dev)
