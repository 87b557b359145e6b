"""Device -> host transfers of the literal drop-in surface at PCIe speed.

The reference hands numpy arrays back (`re_ranking`, reid/rerank.py:127: float16 euclidean_dist + float64 final_dist = 10 N^2 bytes;
`extract_features`, reid/evaluators.py:37-39: a dictionary of CPU tensors).  `Tensor.cpu()` into fresh pageable memory runs at ~10 GB/s
(page faults + a staging copy inside the runtime): 0.25 s for the 2.56 GB of N = 16 000, 45 x the fused grouping leg.  Here the
destination is page-locked memory taken from a pool, so the copy engine writes it directly (PCIe Gen5 x16: ~50 GB/s):

  * `pinned_empty(shape, dtype)` -> a torch CPU tensor over a pinned block of the pool.  A block is handed out again only when no
    tensor, view or numpy array refers to its storage any more (the storage's reference count decides): an array that was handed out
    is never overwritten by a later call (no aliasing), and a loop that drops iteration i's matrices before iteration i + 1
    (selftraining.py's does) re-uses the same pages from the second iteration on -- no page-locking, no page faults.  The FIRST call
    of a size pays the page-locking (hipHostMalloc, a few GB/s).
  * `to_host(t)` -> one asynchronous copy on a copy stream + one wait.
  * `final_dist_to_host(handle)` -> rows of rerank.py:122's float64 matrix produced chunk by chunk on the device (two chunk buffers)
    while the previous chunk travels: the 8 N^2-byte matrix never exists in HBM and the kernel time hides under the copy.
"""
import os
import threading

import torch

_LOCK = threading.Lock()
_BLOCKS = {}          # nbytes -> [(pinned uint8 tensor, use count of its storage while nobody else refers to it)]
_HELD = [0]
_COPY_STREAMS = {}


def _pool_cap():
    return int(float(os.environ.get("SSG_PINNED_POOL_GB", "16")) * (1 << 30))


def _use_count(block):
    return torch._C._storage_Use_Count(block.untyped_storage()._cdata)


def pinned_empty(shape, dtype):
    """uninitialised CPU tensor of `shape` / `dtype` over a page-locked block of the pool (see the module docstring).  A block is
    re-used only while NOTHING but the pool refers to its storage -- tensors, views and numpy arrays made from them all count (the
    storage's own reference count is the test) -- so memory that was handed out is never overwritten behind the caller's back."""
    esz = torch.empty(0, dtype=dtype).element_size()
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = max(n * esz, 1)
    block = None
    with _LOCK:
        for cand, base in _BLOCKS.get(nbytes, ()):
            if _use_count(cand) == base:
                block = cand
                break
    if block is None:
        block = torch.empty(nbytes, dtype=torch.uint8, pin_memory=torch.cuda.is_available())      # (no GPU: the CPU tests of the pool's ownership rule)
        with _LOCK:
            if _HELD[0] + nbytes <= _pool_cap():          # beyond the cap the block is not kept: it is freed with its last user
                _BLOCKS.setdefault(nbytes, []).append((block, _use_count(block)))
                _HELD[0] += nbytes
    return block[: n * esz].view(dtype).view(tuple(int(s) for s in shape))


def pool_stats():
    with _LOCK:
        blocks = [(c, b) for lst in _BLOCKS.values() for c, b in lst]
        return {"blocks": len(blocks), "bytes": _HELD[0], "free_blocks": sum(1 for c, b in blocks if _use_count(c) == b)}


def pool_clear():
    """drop the pool's own references (blocks still in use live on until their users let go)"""
    with _LOCK:
        _BLOCKS.clear(); _HELD[0] = 0


def _copy_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _COPY_STREAMS.get(key)
    if s is None:
        s = _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def to_host(t, wait=True):
    """CUDA tensor -> CPU tensor in pooled page-locked memory: one async copy on the copy stream behind the producer, one wait
    (wait=False: the caller waits later -- `wait_copies` -- e.g. after queueing further copies)"""
    if not t.is_cuda:
        return t
    t = t.contiguous()
    out = pinned_empty(t.shape, t.dtype)
    cs = _copy_stream(t.device)
    cs.wait_stream(torch.cuda.current_stream(t.device))
    with torch.cuda.stream(cs):
        out.copy_(t, non_blocking=True)
    t.record_stream(cs)
    if wait:
        cs.synchronize()
    return out


def wait_copies(dev):
    _copy_stream(dev).synchronize()


def final_dist_to_host(h, chunk_bytes=None):
    """rerank.py:122's float64 final_dist [nrows, N] of a DistHandle straight into pooled page-locked memory: `ssg_final_dist_f64`
    writes chunk k + 1 of the rows into one of two device buffers while chunk k travels over PCIe."""
    from . import _lib
    from ._lib import check, ptr
    h.validate()
    if h.mode != 0:
        return to_host(h.final_dist())
    L = _lib.lib()
    dev, N, nrows = h.device, h.N, h.nrows
    chunk_bytes = int(chunk_bytes or float(os.environ.get("SSG_D2H_CHUNK_MB", "256")) * (1 << 20))
    rows_per = max(1, min(nrows, chunk_bytes // (8 * N)))
    out = pinned_empty((nrows, N), torch.float64)
    main, cs = torch.cuda.current_stream(dev), _copy_stream(dev)
    bufs = [torch.empty((rows_per, N), dtype=torch.float64, device=dev) for _ in range(2 if nrows > rows_per else 1)]
    done = [None, None]                                   # event: the copy out of buffer b has finished
    for k, r0 in enumerate(range(0, nrows, rows_per)):
        nr = min(rows_per, nrows - r0)
        b = k % len(bufs)
        if done[b] is not None:
            main.wait_event(done[b])                      # the kernel may overwrite the buffer only after its previous chunk has left
        check(L.ssg_final_dist_f64(ptr(h.M[r0:]), ptr(h.v), N, h.row0 + r0, nr, h.lambda_value, ptr(bufs[b]), _lib.stream()), "ssg_final_dist_f64")
        ready = torch.cuda.Event(); ready.record(main)
        cs.wait_event(ready)
        with torch.cuda.stream(cs):
            out[r0:r0 + nr].copy_(bufs[b][:nr], non_blocking=True)
            done[b] = torch.cuda.Event(); done[b].record(cs)
    for b_ in bufs:
        b_.record_stream(cs)
    cs.synchronize()
    return out
