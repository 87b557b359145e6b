"""epsilon rule + DBSCAN on MI355X -- host-side mirror of selftraining.py:280-313 and of the
sklearn estimator it instantiates (`DBSCAN(eps, min_samples=4, metric='precomputed', n_jobs=8)`).

`DBSCAN` is a drop-in for the sklearn 1.7.2 class for metric='precomputed': same constructor
arguments, `fit`, `fit_predict`, `labels_`, `core_sample_indices_`, same label numbering
(clusters numbered by their smallest core index; border points take the smallest cluster id
among their core neighbours; noise = -1).  It accepts a numpy/torch matrix, or the
`DistHandle` / `DeviceBackedArray` produced by `ssg_amd.rerank` (no re-upload).
"""
import math
import numbers

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .rerank import DistHandle

_LEVELS = ((51, 12), (39, 12), (27, 12), (15, 12), (3, 12), (0, 3))   # (shift, width) of the radix digits
_MAX_COMPACT = 1 << 24


from .dist import all_reduce_sum as _all_reduce, gather_rows, gather_varlen  # noqa: E402


def as_handle(X, device=None):
    """numpy / torch matrix or handle -> DistHandle on the GPU."""
    if isinstance(X, DistHandle):
        return X
    valid = getattr(X, "valid_handle", None)     # DeviceBackedArray: only the untouched, still read-only original carries one
    h = valid() if callable(valid) else None
    if isinstance(h, DistHandle):
        return h
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if isinstance(X, np.ndarray):
        if X.ndim != 2 or X.shape[0] != X.shape[1]:
            raise ValueError("Precomputed distance matrix must be square, got shape %r" % (X.shape,))
        if X.dtype == np.float16:
            return DistHandle(X.shape[0], 1, torch.from_numpy(np.ascontiguousarray(X)).to(device))
        return DistHandle(X.shape[0], 2, torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(device))
    if torch.is_tensor(X):
        if X.dim() != 2 or X.shape[0] != X.shape[1]:
            raise ValueError("Precomputed distance matrix must be square")
        if X.dtype == torch.float16:
            return DistHandle(X.shape[0], 1, X.to(device).contiguous())
        return DistHandle(X.shape[0], 2, X.to(device=device, dtype=torch.float64).contiguous())
    raise TypeError("unsupported distance matrix type %r" % type(X))


def _eps_finish(L, h, buf, got, n_pow2, top, st):
    """sort the collected keys, numpy pairwise mean of the first `top` -> (eps, key[top-1] as float64)"""
    dev = h.device
    ws_bytes = int(L.ssg_eps_mean_workspace_bytes(top))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    # the summation tables depend on `top` only: uploaded now, while the stream is idle (the caller has just read `top`
    # back), so that sort -> leaves -> tree run back to back without a host wait in between
    check(L.ssg_eps_mean_prepare(top, ptr(ws), ws_bytes, st), "ssg_eps_mean_prepare")
    check(L.ssg_fill_u64(ptr(buf), got, n_pow2, 0xFFFFFFFFFFFFFFFF, st), "ssg_fill_u64")
    check(L.ssg_sort_u64(ptr(buf), n_pow2, st), "ssg_sort_u64")
    check(L.ssg_eps_mean_run(ptr(buf), top, 1 if h.mode == 1 else 0, ptr(ws), ws_bytes, ptr(out), st), "ssg_eps_mean_run")
    res = torch.cat([out, buf[top - 1:top].view(torch.float64)]).cpu().numpy()      # one read-back: eps (+ half bits) and the top-th key
    eps = np.uint16(int(res[1])).view(np.float16) if h.mode == 1 else float(res[0])
    return eps, float(res[2])


def _eps_rule_sampled(L, h, rho, st):
    """Fast path (csrc/cluster.hip "K10, fast path"): sampled float32 threshold, ONE full pass, exactness verified a posteriori.
    Returns (eps, count, top) or None when the verification fails (the caller then runs the radix select)."""
    dev = h.device
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    # strict upper triangle of this row block
    N = h.N
    upper_total = N * (N - 1) // 2
    expected_top = max(int(rho * upper_total), 1)
    stride = max(1, h.nrows // 192)
    hist = torch.zeros(2 * 4097, dtype=torch.int64, device=dev)
    hist1, hist2 = hist[:4097], hist[4097:]
    check(L.ssg_eps_sample_hist(*args, stride, None, ptr(hist1), st), "ssg_eps_sample_hist")
    hist1 = _all_reduce(hist1, h.group)                     # sharded rows: every rank selects from the same global sample
    thr3 = torch.zeros(5, dtype=torch.int64, device=dev)
    check(L.ssg_eps_select_threshold(ptr(hist1), 1.3 * float(rho), ptr(thr3), st), "ssg_eps_select_threshold")
    # second level inside the selected 0.4 %-wide bin (1024 sub-bins): on a dense value distribution one coarse bin holds far
    # more elements than the quantile margin
    check(L.ssg_eps_sample_hist(*args, stride, ptr(thr3), ptr(hist2), st), "ssg_eps_sample_hist")
    hist2 = _all_reduce(hist2, h.group)
    check(L.ssg_eps_refine_threshold(ptr(hist2), ptr(thr3), st), "ssg_eps_refine_threshold")
    cap = max(6 * expected_top * h.nrows // N + (1 << 16), 1 << 16)
    n_cap = max(2048, 1 << (cap - 1).bit_length())
    buf = torch.empty(n_cap, dtype=torch.int64, device=dev)
    cursor = torch.zeros(3, dtype=torch.int64, device=dev)
    sp = getattr(h, "sparse", None) if h.mode == 0 else None
    if sp is not None:
        # row by row through the sparse copy S of J' where the threshold lies below the row's floor (decided on the device; the dense pass
        # for the other rows is queued behind it)
        check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.lambda_value, ptr(thr3), ptr(buf), n_cap, ptr(cursor), ptr(sp["pool"]),
                                        ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), st),
              "ssg_eps_compact_below_s")
    else:
        check(L.ssg_eps_compact_below(*args, ptr(thr3), ptr(buf), n_cap, ptr(cursor), st), "ssg_eps_compact_below")
    pend = h.take_pending() if hasattr(h, "take_pending") else None    # the re-rank's status words ride along with this read-back
    npend = int(pend.numel()) if pend is not None else 0
    local = torch.cat([cursor[:2], thr3[:1]] + ([pend.to(torch.int64)] if pend is not None else []))
    if h.group is not None:
        # sharded rows: ONE flat all-gather of every rank's (candidates, zeros, threshold, status words) and ONE blocking read of
        # the table give each rank its own values, the global sums for the accept / fall-back decision (the same on every rank)
        # and the block lengths of the candidate gather -- no all-reduce + read, no separate size exchange
        import torch.distributed as dist
        table = gather_rows(local.view(1, -1), h.group).tolist()                                                # host round trip 1
        vals = list(table[dist.get_rank(h.group)])
        if pend is not None:
            # a digit overflow / a missed query-expansion guess on ANY rank is seen by every rank (no rank left waiting in a collective)
            vals[3:] = h.global_status(None, table=table, first=3)
        gots = [int(r[0]) for r in table]
        zeros_all, got_all, overflow = sum(int(r[1]) for r in table), sum(gots), int(any(g > n_cap for g in gots))
    else:
        vals = local.tolist()                                                                                    # host round trip 1
    got, zeros, thr_bits = vals[:3]
    if pend is not None:
        if h.resolve_pending(vals[3:3 + npend]):
            # the query expansion had run on too small a guess and the distance matrix was rebuilt just now: the passes above
            # saw the old contents -- run them again (once; the handle has no pending words any more)
            return _eps_rule_sampled(L, h, rho, st)
        h.validate()
    thr = float(np.uint32(thr_bits & 0xFFFFFFFF).view(np.float32))
    if h.group is None:
        zeros_all, got_all, overflow = int(zeros), int(got), int(got > n_cap)
    count = upper_total - zeros_all
    top = int(np.round(rho * count))                  # np.round: half to even (selftraining.py:292)
    if top <= 0:
        return (np.float16(np.nan) if h.mode == 1 else float("nan")), count, top
    if overflow or got_all < top or not np.isfinite(thr):
        return None
    if h.group is not None:
        from .dist import gather_ragged
        allk = gather_ragged(buf[:got], gots, h.group)
        got = int(allk.shape[0])
        n_pow2 = max(2048, 1 << (got - 1).bit_length())
        buf = torch.empty(n_pow2, dtype=torch.int64, device=dev)
        buf[:got] = allk
    else:
        n_pow2 = max(2048, 1 << (max(got, 1) - 1).bit_length())
    eps, key_top = _eps_finish(L, h, buf, got, n_pow2, top, st)          # host round trip 2
    # every element that was NOT collected has surrogate >= thr, hence exact value >= thr - err: the `top` smallest are all
    # among the collected keys iff the top-th of them is below that
    if not (key_top < thr - 1e-6 * (1.0 + abs(thr))):
        return None
    return eps, count, top


def eps_rule(X, rho):
    """selftraining.py:289-293 on device.

    eps = mean of the round(rho*count) smallest non-zero entries of the strict upper triangle.
    float64 views (mode 0/2) -> python float, bit-identical to numpy's pairwise-sum mean;
    half matrix (mode 1) -> np.float16 like `tri_mat[:top].mean()` on a float16 array.
    Returns (eps, count, top_num).
    """
    import os
    L = _lib.lib()
    h = as_handle(X)
    dev, st = h.device, stream()
    if os.environ.get("SSG_EPS_PATH", "sampled") == "sampled" and float(rho) > 0 and h.N >= 64:
        r = _eps_rule_sampled(L, h, float(rho), st)      # (validates the handle with its first read-back)
        if r is not None:
            return r
    h.validate()
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    prefix, below, count, top = 0, 0, None, None
    key_max = None
    for shift, width in _LEVELS:
        hist = torch.zeros(4097, dtype=torch.int64, device=dev)
        check(L.ssg_eps_hist(*args, prefix, shift, width, 1 if count is None else 0, ptr(hist), st), "ssg_eps_hist")
        hist = _all_reduce(hist, h.group).cpu().numpy()
        if count is None:
            count = int(hist[4096])
            top = int(np.round(rho * count))          # np.round: half to even (selftraining.py:292)
            if top <= 0:
                # numpy: mean of an empty slice -> nan (+ RuntimeWarning); sklearn then rejects eps
                return (np.float16(np.nan) if h.mode == 1 else float("nan")), count, top
        cum = below + np.cumsum(hist[: 1 << width])
        b = int(np.searchsorted(cum, top, side="left"))   # first digit whose cumulative count reaches top
        ncand = int(cum[b])
        if ncand <= _MAX_COMPACT or shift == 0:
            key_max = (((prefix << width) | b) << shift) | ((1 << shift) - 1)
            break
        below = int(cum[b - 1]) if b > 0 else below
        prefix = (prefix << width) | b
    # compact every key <= key_max, sort, pairwise mean of the first `top`
    n_pow2 = max(2048, 1 << (ncand - 1).bit_length())
    buf = torch.empty(n_pow2, dtype=torch.int64, device=dev)
    cursor = torch.zeros(1, dtype=torch.int64, device=dev)
    check(L.ssg_eps_compact(*args, key_max, ptr(buf), n_pow2, ptr(cursor), st), "ssg_eps_compact")
    got = int(cursor.item())
    if h.group is not None:
        # sharded rows: gather the (small) candidate sets of every rank
        allk = gather_varlen(buf[:got], h.group)
        got = int(allk.shape[0])
        buf[:got] = allk
    if got != ncand:
        raise _lib.SSGError("eps_rule: compaction found %d keys, histogram promised %d" % (got, ncand))
    eps, _ = _eps_finish(L, h, buf, got, n_pow2, top, st)
    return eps, count, top


# (top, device index, stream) -> workspace with the pairwise-summation tables of `top` summands already uploaded (they depend on top
# only).  The workspace also holds the node values of a run: calls on two streams must not share it, hence the stream in the key.
_EPS_TREES = {}


def _eps_tree(L, top, dev, st):
    key = (int(top), dev.index, int(getattr(st, "value", 0) or 0))
    ws = _EPS_TREES.get(key)
    if ws is None:
        ws_bytes = int(L.ssg_eps_mean_workspace_bytes(top))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(L.ssg_eps_mean_prepare(top, ptr(ws), ws_bytes, st), "ssg_eps_mean_prepare")      # blocks on the stream: once per (top, device)
        if len(_EPS_TREES) >= 8:
            _EPS_TREES.pop(next(iter(_EPS_TREES)))
        _EPS_TREES[key] = ws
    return ws


def sparse_row_split(X, rho, eps=None):
    """Diagnostic (bench.py / tools/run_configs.py, untimed): how many rows of a re-rank handle the eps rule's one full pass and the region
    query walk through the sparse copy S and how many they hand to the dense pass -> {'eps': (rows_sparse, rows_dense), 'region': (...)}.
    Runs the two passes once more into scratch buffers and reads their row masks (a blocking read each); None without a sparse copy."""
    L = _lib.lib()
    h = as_handle(X)
    sp = getattr(h, "sparse", None) if h.mode == 0 else None
    if sp is None:
        return None
    h.validate()
    dev, st, N = h.device, stream(), h.N
    if not h.sparse_complete():
        return {"eps": (0, h.nrows), "region": (0, h.nrows)}
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    z = torch.zeros(2 * 4097 + 5 + 1 + 3 + 2, dtype=torch.int64, device=dev)
    thr3, cursor, ecur = z[8194:8199], z[8200:8203], z[8203:8205]
    check(L.ssg_eps_sample_threshold(*args, max(1, h.nrows // 192), 1.3 * float(rho), ptr(z[:8194]), ptr(thr3), ptr(z[8199:8200]), None, st), "ssg_eps_sample_threshold")
    n_cap = 1 << 16
    buf = torch.empty(n_cap, dtype=torch.int64, device=dev)
    mask = torch.zeros(h.nrows, dtype=torch.uint8, device=dev)
    check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.lambda_value, ptr(thr3), ptr(buf), n_cap, ptr(cursor), ptr(sp["pool"]),
                                    ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(mask), st), "ssg_eps_compact_below_s")
    dense_e = int(mask.ne(0).sum().item())
    out = {"eps": (h.nrows - dense_e, dense_e)}
    if eps is not None:
        cnt = torch.empty(h.nrows, dtype=torch.int32, device=dev)
        edges = torch.empty((1 << 16, 2), dtype=torch.int32, device=dev)
        mask.zero_()
        check(L.ssg_region_query_s(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.lambda_value, float(eps), ptr(sp["pool"]), ptr(sp["seg_off"]), ptr(sp["seg_len"]),
                                   sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(mask), ptr(cnt), ptr(edges), 1 << 16, ptr(ecur), st), "ssg_region_query_s")
        dense_r = int(mask.ne(0).sum().item())
        out["region"] = (h.nrows - dense_r, dense_r)
    return out


def _triangle_share(N, lo, hi):
    """fraction of the strict upper triangle of an N x N matrix that lies in rows [lo, hi)"""
    tot = N * (N - 1) // 2
    part = sum(N - 1 - i for i in (lo, hi - 1)) * (hi - lo) // 2 if hi > lo else 0       # arithmetic series
    return part / max(tot, 1)


def _eps_rule_dbscan_sharded(L, h, rho, min_samples, two_calls):
    """`eps_rule_dbscan` on row-sharded handles (round 6): the same device chain with TWO collectives and ONE blocking read per split
    (the two-call form: five collectives -- two histogram all-reduces, the status table, the candidates, the edge table -- and four reads).

      1. every rank samples ITS OWN rows and picks its own float32 threshold at the 1.3 * rho quantile (no histogram all-reduce: the
         a-posteriori check below only needs every uncollected key to lie above the SMALLEST of the ranks' thresholds), compacts the
         keys below it into a buffer of the capacity all ranks share;
      2. collective 1: the fixed-capacity buffers + (count, zeros) + threshold of every rank in one flat all-gather; every rank
         concatenates them on the device (`ssg_concat_segments_u64`: the counts never reach the host), sorts, sums numpy's pairwise tree
         and runs `ssg_eps_check` against the minimum threshold -- identical inputs, identical eps on every rank;
      3. region query of the local rows with eps read from device memory;
      4. collective 2: edge count + neighbour counts + fixed-capacity edge list + the re-rank's status words of every rank; concatenated
         on the device, components and labels on every rank (redundant: N-sized, ~20 us);
      5. THE read: labels, neighbour counts, eps, check words, every rank's status words.
    Any failed check is seen identically by every rank (all of them decide from the same gathered words) and the two-call form answers."""
    import torch.distributed as dist
    from .dist import gather_packed, shard_bounds
    import os
    g = h.group
    world, rk = dist.get_world_size(g), dist.get_rank(g)
    dev, st, N = h.device, stream(), h.N
    upper_total = N * (N - 1) // 2
    top_guess = int(np.round(rho * upper_total))
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    stride = max(1, h.nrows // 192)
    z = torch.zeros(2 * 4097 + 5 + 3 + 6 + 2 + 2 + 1 + 2 + 2 + 2 + 1 + 1, dtype=torch.int64, device=dev)
    hist1, tick = z[:8194], z[8220:8221]                           # (both levels' histograms back to back; two uint32 tickets)
    thr3, cursor, status6, ecur, eps2 = z[8194:8199], z[8199:8202], z[8202:8208], z[8208:8210], z[8210:8212].view(torch.float64)
    sort_fail, ktot, gcur, etot, thrg = z[8212:8213], z[8213:8215], z[8215:8217], z[8217:8219], z[8219:8220]
    # the LOCAL quantile is taken wider than the one-GPU chain's 1.3 * rho: a rank whose rows hold more small distances than the average
    # (big identities) would otherwise cut below the global top-th key and fail the check -- 1.6 tolerates a 60 % denser block
    qf = float(os.environ.get("SSG_EPS_SHARD_QUANTILE", "1.6"))
    check(L.ssg_eps_sample_threshold(*args, stride, qf * rho, ptr(hist1), ptr(thr3), ptr(tick), None, st), "ssg_eps_sample_threshold")
    # one capacity for every rank (the buffers travel in a flat all-gather): twice the expected candidates of the rank with the largest
    # part of the triangle (rank 0: 15/64 of it on 8 ranks, not 1/8) + a floor
    share = max(_triangle_share(N, *shard_bounds(N, r, world)) for r in range(world))
    n_cap = int(2 * qf * top_guess * share) + (1 << 16)
    buf = torch.empty(n_cap, dtype=torch.int64, device=dev)
    sp = getattr(h, "sparse", None) if h.mode == 0 else None
    if sp is not None:
        check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.lambda_value, ptr(thr3), ptr(buf), n_cap, ptr(cursor), ptr(sp["pool"]),
                                        ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), st),
              "ssg_eps_compact_below_s")
    else:
        check(L.ssg_eps_compact_below(*args, ptr(thr3), ptr(buf), n_cap, ptr(cursor), st), "ssg_eps_compact_below")
    # ---- collective 1: candidates + (count, zeros, threshold) of every rank
    g_head, g_buf = gather_packed([torch.cat([cursor[:2], thr3[:1]]), buf], g)          # [world, 3], [world, n_cap] (strided views of the receive buffer)
    cap_all = world * n_cap
    n_pow2 = max(2048, 1 << (cap_all - 1).bit_length())
    allk = torch.empty(n_pow2, dtype=torch.int64, device=dev)
    check(L.ssg_concat_segments_u64(ptr(g_buf), world, n_cap, g_buf.stride(0), ptr(g_head), g_head.stride(0), ptr(allk), ptr(ktot), st), "ssg_concat_segments_u64")
    # the words the check reads: candidates (pushed past the capacity when a rank's buffer overflowed), zeros of the whole triangle,
    # the smallest threshold (positive float32 bit patterns order like the integers)
    gcur[0:1] = ktot[0:1] + ktot[1:2] * (1 << 62)
    gcur[1:2] = g_head[:, 1].sum()
    thrg[0:1] = g_head[:, 2].min()
    tree = _eps_tree(L, top_guess, dev, st)
    if os.environ.get("SSG_EPS_SORT", "sample") == "bitonic" or qf * top_guess > 2.4e7 or 4.0e5 < qf * top_guess <= 4.0e6:
        check(L.ssg_sort_u64_dev(ptr(allk), n_pow2, ptr(ktot), st), "ssg_sort_u64_dev")
        sf = None
    elif qf * top_guess > 4.0e6:
        sws_bytes = int(L.ssg_samplesort_u64_big_workspace_bytes(n_pow2))
        sws = torch.empty(sws_bytes, dtype=torch.uint8, device=dev)
        check(L.ssg_samplesort_u64_big_dev(ptr(allk), n_pow2, ptr(ktot), ptr(sws), sws_bytes, ptr(sort_fail), st), "ssg_samplesort_u64_big_dev")
        sf = sort_fail
    else:
        sws_bytes = int(L.ssg_samplesort_u64_workspace_bytes(n_pow2))
        sws = torch.empty(sws_bytes, dtype=torch.uint8, device=dev)
        check(L.ssg_samplesort_u64_dev(ptr(allk), n_pow2, ptr(ktot), ptr(sws), sws_bytes, ptr(sort_fail), st), "ssg_samplesort_u64_dev")
        sf = sort_fail
    check(L.ssg_eps_mean_check(ptr(allk), top_guess, 1 if h.mode == 1 else 0, ptr(tree), tree.numel(), ptr(eps2), ptr(gcur), ptr(thrg), rho, upper_total, cap_all,
                               ptr(status6), ptr(sf), st), "ssg_eps_mean_check")
    # ---- region query of the local rows with eps read from the device
    mx_rows = -(-N // world)
    cnt = torch.zeros(mx_rows, dtype=torch.int32, device=dev)                             # (padded to the longest block: one shape on every rank)
    ecap = max(64 * mx_rows, 1 << 16)
    edges = torch.empty((ecap, 2), dtype=torch.int32, device=dev)
    if sp is not None:
        check(L.ssg_region_query_s_dev(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.lambda_value, ptr(eps2), ptr(sp["pool"]), ptr(sp["seg_off"]), ptr(sp["seg_len"]),
                                       sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), ptr(cnt), ptr(edges), ecap, ptr(ecur), st),
              "ssg_region_query_s_dev")
    else:
        check(L.ssg_region_query_dev(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.mode, h.lambda_value, ptr(eps2), ptr(cnt), ptr(edges), ecap, ptr(ecur), st),
              "ssg_region_query_dev")
    # ---- collective 2: edge count, neighbour counts, edge list and the re-rank's status words of every rank
    pend = h.take_pending() if hasattr(h, "take_pending") else None
    npend = int(pend.numel()) if pend is not None else 0
    words = torch.cat([ecur[:1]] + ([pend.to(torch.int64)] if pend is not None else []))
    g_words, g_cnt, g_edges = gather_packed([words, cnt, edges.view(torch.int64).view(-1)], g)    # [world, 1 + npend], [world, mx_rows], [world, ecap]
    edges_all = torch.empty(world * ecap, dtype=torch.int64, device=dev)
    check(L.ssg_concat_segments_u64(ptr(g_edges), world, ecap, g_edges.stride(0), ptr(g_words), g_words.stride(0), ptr(edges_all), ptr(etot), st),
          "ssg_concat_segments_u64")
    rows = [shard_bounds(N, r, world) for r in range(world)]
    cnt_all = torch.cat([g_cnt[r, :hi - lo] for r, (lo, hi) in enumerate(rows)])
    ws_bytes = int(L.ssg_dbscan_cc_workspace_bytes(N))
    ws_buf = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    labels = torch.empty(N, dtype=torch.int64, device=dev)
    check(L.ssg_dbscan_cc_dev(ptr(cnt_all), ptr(edges_all), ptr(etot), world * ecap, N, int(min_samples), ptr(ws_buf), ws_bytes, ptr(labels), st), "ssg_dbscan_cc_dev")
    host = torch.cat([etot, status6, eps2.view(torch.int64), g_words.reshape(-1), labels, cnt_all.to(torch.int64)]).cpu().numpy()   # THE read
    e_over, ok, zeros, top = int(host[1]), int(host[2]), int(host[4]), int(host[5])
    eps_f64, eps_hbits = float(host[8:9].view(np.float64)[0]), int(host[9:10].view(np.float64)[0]) if ok else 0
    o = 10
    table = host[o:o + world * (1 + npend)].reshape(world, 1 + npend)
    o += world * (1 + npend)
    if pend is not None:
        vals = h.global_status(None, table=[[int(x) for x in r] for r in table], first=1)
        if h.resolve_pending(vals):
            return eps_rule_dbscan(h, rho, min_samples)      # the query expansion had run on too small a guess: the matrix was rebuilt (by every rank), run again (once)
        h.validate()
    if not ok:
        return two_calls()                                    # zeros in the triangle, a full buffer, or a sample that missed: the two-call path decides (exactly)
    count = upper_total - zeros
    eps = np.uint16(eps_hbits).view(np.float16) if h.mode == 1 else eps_f64
    if e_over:                                                # an edge list was too small on some rank: eps is known, the two-call region query sizes it exactly
        est = DBSCAN(eps=eps, min_samples=min_samples, metric="precomputed").fit(h)
        return eps, count, top, est.labels_, est.core_sample_indices_
    lab = host[o:o + N].copy()
    core = np.nonzero(host[o + N:o + 2 * N] >= int(min_samples))[0]
    return eps, count, top, lab, core


def eps_rule_dbscan(X, rho, min_samples=4):
    """selftraining.py:289-306 in ONE device-resident chain (round 5): eps rule -> region query -> connected components with eps, the
    candidate count and the edge count left on the device, and ONE blocking read at the end (labels, neighbour counts, eps, the check
    words and the re-rank's status words).  Returns (eps, count, top, labels, core_sample_indices) -- exactly what
    `eps_rule(X, rho)` followed by `DBSCAN(eps, min_samples, metric='precomputed').fit(X)` returns; that two-call form stays the
    API (and the fallback of every case this chain does not cover: the radix-select path, a failed check).  Sharded rows run the same
    chain with two collectives (`_eps_rule_dbscan_sharded`).

    What the host decides BEFORE the data is seen, and the device verifies: the number of summands top = round(rho * count) assumes no
    zero entry in the strict upper triangle (count = N(N-1)/2; a zero -- duplicate images -- makes `ssg_eps_check` fail and the
    two-call path runs instead); the sort runs on the device's own candidate count inside a buffer sized by the sampling bound."""
    import os
    L = _lib.lib()
    h = as_handle(X)
    rho = float(rho)
    if not isinstance(min_samples, numbers.Integral) or min_samples < 1:
        raise ValueError("The 'min_samples' parameter of DBSCAN must be an int in the range [1, inf). Got %r instead." % (min_samples,))

    def two_calls():
        eps, count, top = eps_rule(h, rho)
        est = DBSCAN(eps=eps, min_samples=min_samples, metric="precomputed").fit(h)
        return eps, count, top, est.labels_, est.core_sample_indices_

    N = h.N
    upper_total = N * (N - 1) // 2
    top_guess = int(np.round(rho * upper_total))
    if (os.environ.get("SSG_EPS_PATH", "sampled") != "sampled" or os.environ.get("SSG_EPS_FUSED", "1") == "0" or not rho > 0
            or N < 64 or top_guess <= 0):
        return two_calls()
    if h.group is not None:
        return _eps_rule_dbscan_sharded(L, h, rho, min_samples, two_calls)
    dev, st = h.device, stream()
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    # ---- ONE zero-filled allocation holds every small table of the chain AND everything the host reads at the end (round 6: the read-back
    # needs no concatenation / conversion launches, the components' workspace no initialisation launch of its own), int64 words:
    #   [hist1 4097 | hist2 4097 | tickets 1 | thr5 5 | cursor 3 | sort_fail 1 | splitters 1024 | bucket counts 1024 || ecur 2 | status6 6 | eps2 2 |
    #    cnt (int32) | ws: parent, lab (int32), ...]
    stride = max(1, h.nrows // 192)
    ws_bytes = int(L.ssg_dbscan_cc_workspace_bytes(N))
    ncnt = (h.nrows + 1) // 2
    o_small, o_out = 2 * 4097, 2 * 4097 + 10 + 2048
    o_cnt = o_out + 10
    o_ws = o_cnt + ncnt
    z = torch.zeros(o_ws + (ws_bytes + 7) // 8, dtype=torch.int64, device=dev)
    hist = z[:o_small]
    tickets, thr3, cursor, sort_fail = z[o_small:o_small + 1], z[o_small + 1:o_small + 6], z[o_small + 6:o_small + 9], z[o_small + 9:o_small + 10]
    splitters, gcount = z[o_small + 10:o_small + 1034], z[o_small + 1034:o_small + 2058]
    ecur, status6, eps2 = z[o_out:o_out + 2], z[o_out + 2:o_out + 8], z[o_out + 8:o_out + 10].view(torch.float64)
    cnt = z[o_cnt:o_ws].view(torch.int32)[:h.nrows]
    ws_buf = z[o_ws:].view(torch.uint8)
    # ---- the sampled threshold (two launches: each level's last workgroup selects) and the one full pass
    fused = os.environ.get("SSG_EPS_FUSED_LAUNCHES", "1") != "0"      # (0: the separate selection / check launches of round 5 -- A/B switch)
    if fused:
        # (the interpolated sort splitters are only computed for the rejected SSG_EPS_PRESPLIT experiment: one thread may own hundreds of them,
        # 33 us in the selecting workgroup -- round 6 measured the chain at 0.39 instead of 0.31 ms while they were always written)
        presplit = os.environ.get("SSG_EPS_PRESPLIT", "0") == "1"
        check(L.ssg_eps_sample_threshold(*args, stride, 1.3 * rho, ptr(hist), ptr(thr3), ptr(tickets), ptr(splitters) if presplit else None, st),
              "ssg_eps_sample_threshold")
    else:
        check(L.ssg_eps_sample_hist(*args, stride, None, ptr(hist[:4097]), st), "ssg_eps_sample_hist")
        check(L.ssg_eps_select_threshold(ptr(hist[:4097]), 1.3 * rho, ptr(thr3), st), "ssg_eps_select_threshold")
        check(L.ssg_eps_sample_hist(*args, stride, ptr(thr3), ptr(hist[4097:]), st), "ssg_eps_sample_hist")
        check(L.ssg_eps_refine_threshold(ptr(hist[4097:]), ptr(thr3), st), "ssg_eps_refine_threshold")
    # the threshold sits at the 1.3 * rho quantile of ~3 M sampled entries (about 1 % sampling error on the count below it): twice the
    # expected top + a floor holds the candidates with a wide margin -- and keeps the launched sort network two levels shorter than the
    # two-call path's 6x bound (a fuller buffer fails the check and the two-call path answers)
    cap = max(2 * top_guess * h.nrows // N + (1 << 16), 1 << 16)
    n_cap = max(2048, 1 << (cap - 1).bit_length())
    buf = torch.empty(n_cap, dtype=torch.int64, device=dev)
    sp = getattr(h, "sparse", None) if h.mode == 0 else None
    if sp is not None:
        check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.lambda_value, ptr(thr3), ptr(buf), n_cap, ptr(cursor), ptr(sp["pool"]),
                                        ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), st),
              "ssg_eps_compact_below_s")
    else:
        check(L.ssg_eps_compact_below(*args, ptr(thr3), ptr(buf), n_cap, ptr(cursor), st), "ssg_eps_compact_below")
    # ---- sort (device-sized), numpy's pairwise mean of the first top_guess keys + the checks in the tree kernel -- no read-back
    tree = _eps_tree(L, top_guess, dev, st)
    # sample sort (4 launches) while its 1024 sorting buckets stay LDS-sized: ~1.3 * top candidates expected, a bucket may run to 4x the
    # mean, 2048 keys fit -> up to 4e5 candidates (N = 16 000: 2.7e5); above that the bitonic network (25+ launches) as in round 4
    expect = 1.3 * top_guess * h.nrows / N
    if os.environ.get("SSG_EPS_SORT", "sample") == "bitonic" or expect > 2.4e7 or 4.0e5 < expect <= 4.0e6:
        check(L.ssg_sort_u64_dev(ptr(buf), n_cap, ptr(cursor), st), "ssg_sort_u64_dev")
        sf = None
    elif expect > 4.0e6:
        # round 6: 4095 splitters, buckets of up to 16 384 keys sorted in 128 KB of LDS (N = 128 000: 1.7e7 candidates: 3.0 ms against 6.9 of the
        # network; measured slower than the network below ~4e6 keys -- tools/time_sort.py -- so N = 30 000's 9.4e5 candidates keep the network)
        sws_bytes = int(L.ssg_samplesort_u64_big_workspace_bytes(n_cap))
        sws = torch.empty(sws_bytes, dtype=torch.uint8, device=dev)
        check(L.ssg_samplesort_u64_big_dev(ptr(buf), n_cap, ptr(cursor), ptr(sws), sws_bytes, ptr(sort_fail), st), "ssg_samplesort_u64_big_dev")
        sf = sort_fail
    else:
        sws_bytes = int(L.ssg_samplesort_u64_workspace_bytes(n_cap))
        sws = torch.empty(sws_bytes, dtype=torch.uint8, device=dev)
        if fused and os.environ.get("SSG_EPS_PRESPLIT", "0") == "1":
            # (opt-in, measured and rejected: splitters interpolated in the sample histogram save the 37 us sample sort, but final_dist takes
            # few distinct values -- half J' plus a half source term -- and thousands of EQUAL keys then share one sorting bucket: the
            # sorted sample gives such a value a bucket of its own, an interpolated splitter cannot.  20 ms instead of 0.33 at N = 16 000.)
            check(L.ssg_samplesort_u64_presplit_dev(ptr(buf), n_cap, ptr(cursor), ptr(splitters), ptr(gcount), ptr(sws), sws_bytes, ptr(sort_fail), st),
                  "ssg_samplesort_u64_presplit_dev")
        else:
            check(L.ssg_samplesort_u64_dev(ptr(buf), n_cap, ptr(cursor), ptr(sws), sws_bytes, ptr(sort_fail), st), "ssg_samplesort_u64_dev")
        sf = sort_fail
    if fused:
        check(L.ssg_eps_mean_check(ptr(buf), top_guess, 1 if h.mode == 1 else 0, ptr(tree), tree.numel(), ptr(eps2), ptr(cursor), ptr(thr3), rho, upper_total, n_cap,
                                   ptr(status6), ptr(sf), st), "ssg_eps_mean_check")
    else:
        check(L.ssg_eps_mean_run(ptr(buf), top_guess, 1 if h.mode == 1 else 0, ptr(tree), tree.numel(), ptr(eps2), st), "ssg_eps_mean_run")
        check(L.ssg_eps_check(ptr(buf), ptr(cursor), ptr(thr3), rho, upper_total, top_guess, n_cap, ptr(eps2), ptr(status6), ptr(sf), st), "ssg_eps_check")
    # ---- region query with eps read from the device, components; the labels stay int32 in the workspace (0x7fffffff = noise)
    ecap = max(64 * h.nrows, 1 << 16)
    edges = torch.empty((ecap, 2), dtype=torch.int32, device=dev)
    if sp is not None:
        check(L.ssg_region_query_s_dev(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.lambda_value, ptr(eps2), ptr(sp["pool"]), ptr(sp["seg_off"]), ptr(sp["seg_len"]),
                                       sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), ptr(cnt), ptr(edges), ecap, ptr(ecur), st),
              "ssg_region_query_s_dev")
    else:
        check(L.ssg_region_query_dev(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.mode, h.lambda_value, ptr(eps2), ptr(cnt), ptr(edges), ecap, ptr(ecur), st),
              "ssg_region_query_dev")
    check(L.ssg_dbscan_cc_dev(ptr(cnt), ptr(edges), ptr(ecur), ecap, N, int(min_samples), ptr(ws_buf), ws_bytes, None, st), "ssg_dbscan_cc_dev")
    # ---- THE read: the contiguous tail of `z` up to the end of the int32 labels, and the re-rank's raw status words, as DMA copies into one
    # page-locked buffer behind the chain -- no concatenation launch -- and one wait
    from . import hostio
    pieces = h.pending_pieces() if hasattr(h, "pending_pieces") else None
    n_out = (o_ws - o_out) + (2 * N * 4 + 7) // 8                    # ecur .. cnt, then parent[N] + lab[N] of the workspace
    extra = [p_.reshape(-1).view(torch.uint8) for p_ in (pieces or [])]
    hostbuf = hostio.pinned_empty((n_out * 8 + sum((e_.numel() + 7) // 8 * 8 for e_ in extra),), torch.uint8)
    hostbuf[:n_out * 8].copy_(z[o_out:o_out + n_out].view(torch.uint8), non_blocking=True)
    off_b = n_out * 8
    for e_ in extra:
        hostbuf[off_b:off_b + e_.numel()].copy_(e_, non_blocking=True)
        off_b += (e_.numel() + 7) // 8 * 8
    torch.cuda.current_stream(dev).synchronize()
    raw = hostbuf.numpy()
    host = raw[:n_out * 8].view(np.int64)
    ne, ok, got, zeros, top = int(host[0]), int(host[2]), int(host[3]), int(host[4]), int(host[5])
    eps_f64, eps_hbits = float(host[8:9].view(np.float64)[0]), int(host[9:10].view(np.float64)[0]) if ok else 0
    if pieces is not None:
        vals, off_b = [], n_out * 8
        for p_, e_ in zip(pieces, extra):
            vals += [int(x) for x in raw[off_b:off_b + e_.numel()].view({torch.int32: np.int32, torch.int64: np.int64}[p_.dtype])]
            off_b += (e_.numel() + 7) // 8 * 8
        if h.resolve_pending(vals):
            return eps_rule_dbscan(h, rho, min_samples)      # the query expansion had run on too small a guess: the matrix was rebuilt, run again (once)
        h.validate()
    if not ok:
        return two_calls()                                    # zeros in the triangle, or the sample missed: the two-call path decides (exactly)
    count = upper_total - zeros
    eps = np.uint16(eps_hbits).view(np.float16) if h.mode == 1 else eps_f64
    if ne > ecap:                                             # the edge list was too small: eps is known now, the region query is redone with the exact size
        est = DBSCAN(eps=eps, min_samples=min_samples, metric="precomputed").fit(h)
        return eps, count, top, est.labels_, est.core_sample_indices_
    cnt_h = raw[(o_cnt - o_out) * 8:(o_cnt - o_out) * 8 + 4 * h.nrows].view(np.int32)
    lab32 = raw[(o_ws - o_out) * 8 + 4 * N:(o_ws - o_out) * 8 + 8 * N].view(np.int32)
    lab = lab32.astype(np.int64)
    lab[lab32 == 0x7fffffff] = -1
    core = np.nonzero(cnt_h >= int(min_samples))[0]
    return eps, count, top, lab, core


class DBSCAN:
    """sklearn.cluster.DBSCAN look-alike for metric='precomputed' running on the GPU."""

    def __init__(self, eps=0.5, *, min_samples=5, metric="euclidean", metric_params=None, algorithm="auto", leaf_size=30, p=None,
                 n_jobs=None):
        self.eps, self.min_samples, self.metric, self.metric_params = eps, min_samples, metric, metric_params
        self.algorithm, self.leaf_size, self.p, self.n_jobs = algorithm, leaf_size, p, n_jobs

    def _validate(self):
        # sklearn raises InvalidParameterError (a ValueError) for these
        if not isinstance(self.eps, numbers.Real) or isinstance(self.eps, bool) or not (float(self.eps) > 0.0) or math.isnan(float(self.eps)):
            raise ValueError("The 'eps' parameter of DBSCAN must be a float in the range (0.0, inf). Got %r instead." % (self.eps,))
        if not isinstance(self.min_samples, numbers.Integral) or self.min_samples < 1:
            raise ValueError("The 'min_samples' parameter of DBSCAN must be an int in the range [1, inf). Got %r instead." % (self.min_samples,))
        if self.metric != "precomputed":
            raise ValueError("ssg_amd.cluster.DBSCAN implements metric='precomputed' only (the SSG grouping path, "
                             "selftraining.py:295); got metric=%r" % (self.metric,))

    def fit(self, X, y=None, sample_weight=None):
        self._validate()
        if sample_weight is not None:
            raise ValueError("sample_weight is not supported on the precomputed GPU path")
        L = _lib.lib()
        h = as_handle(X)
        dev, st, N = h.device, stream(), h.N
        eps = float(self.eps)
        cnt = torch.empty(h.nrows, dtype=torch.int32, device=dev)
        mx_rows = h.nrows
        if h.group is not None:
            import torch.distributed as dist
            mx_rows = -(-N // dist.get_world_size(h.group))      # the longest row block: every rank sizes its edge list alike (they travel in one flat gather)
        cap = max(64 * mx_rows, 1 << 16)
        ws_bytes = int(L.ssg_dbscan_cc_workspace_bytes(N))
        ws_buf = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        labels = torch.empty(N, dtype=torch.int64, device=dev)
        if hasattr(h, "validate"):
            h.validate()
        while True:
            edges = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            cursor = torch.zeros(1, dtype=torch.int64, device=dev)
            sp = getattr(h, "sparse", None) if h.mode == 0 else None
            if sp is not None:
                # row by row through the sparse copy S of J' where eps lies below the row's floor (decided on the device), dense scan of the rest
                cursor = torch.zeros(2, dtype=torch.int64, device=dev)
                check(L.ssg_region_query_s(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.lambda_value, eps, ptr(sp["pool"]), ptr(sp["seg_off"]), ptr(sp["seg_len"]),
                                           sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), ptr(cnt), ptr(edges), cap, ptr(cursor), st),
                      "ssg_region_query_s")
                cursor = cursor[:1]
            else:
                check(L.ssg_region_query(ptr(h.M), ptr(h.v), N, h.row0, h.nrows, h.mode, h.lambda_value, eps, ptr(cnt), ptr(edges), cap,
                                         ptr(cursor), st), "ssg_region_query")
            if h.group is None:
                # one GPU: components and labels follow on the stream, the edge count stays on the device; ONE read-back at
                # the end brings labels, neighbour counts and the count (which tells whether the edge list was big enough)
                check(L.ssg_dbscan_cc_dev(ptr(cnt), ptr(edges), ptr(cursor), cap, N, int(self.min_samples), ptr(ws_buf), ws_bytes, ptr(labels), st),
                      "ssg_dbscan_cc_dev")
                host = torch.cat([cursor, labels, cnt.to(torch.int64)]).cpu().numpy()
                ne = int(host[0])
                if ne <= cap:
                    self.labels_ = host[1:1 + N].copy()
                    self.core_sample_indices_ = np.nonzero(host[1 + N:] >= int(self.min_samples))[0]
                    break
                cap = ne       # the cursor counted every hit: retry once with the exact size
                continue
            # sharded rows: ONE flat all-gather carries every rank's edge count, neighbour counts and edge list (at the common capacity:
            # 64 edges per row is 1 MB per rank at N = 16 000 over 8 ranks) and ONE blocking read of the counts tells every rank whether
            # the lists were big enough and where each rank's edges end; the labels come back with one more read (round 4: two
            # gathers -- counts, then the exact-size lists -- plus the neighbour-count gather)
            from .dist import gather_packed, shard_bounds
            import torch.distributed as dist
            ws = dist.get_world_size(h.group)
            cnt_pad = cnt if h.nrows == mx_rows else torch.cat([cnt, torch.zeros(mx_rows - h.nrows, dtype=cnt.dtype, device=dev)])
            g_cur, g_cnt, g_edges = gather_packed([cursor.view(1), cnt_pad, edges], h.group)
            nes = [int(x) for x in g_cur.flatten().tolist()]
            if max(nes) <= cap:
                rows = [shard_bounds(N, r, ws) for r in range(ws)]
                cnt_all = torch.cat([g_cnt[r, :hi - lo] for r, (lo, hi) in enumerate(rows)])
                edges = torch.cat([g_edges[r, :nes[r]] for r in range(ws)], dim=0).contiguous()
                ne = int(edges.shape[0])
                check(L.ssg_dbscan_cc(ptr(cnt_all), ptr(edges), ne, N, int(self.min_samples), ptr(ws_buf), ws_bytes, ptr(labels), st), "ssg_dbscan_cc")
                host = torch.cat([labels, cnt_all.to(torch.int64)]).cpu().numpy()
                self.labels_ = host[:N].copy()
                self.core_sample_indices_ = np.nonzero(host[N:] >= int(self.min_samples))[0]
                break
            cap = max(nes)          # every rank retries with the same capacity (the collectives stay matched)
        self.n_features_in_ = N
        return self

    def fit_predict(self, X, y=None, sample_weight=None):
        return self.fit(X, sample_weight=sample_weight).labels_
