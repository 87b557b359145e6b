"""k-reciprocal re-ranking on MI355X -- host-side mirror of reid/rerank.py.

`re_ranking` keeps the reference signature and return types (reid/rerank.py:27,66,127):
numpy float32 features in, `(euclidean_dist float16 [N,N], final_dist float64 [N,N] | None)`
out, three progress prints.  The work runs as hand-written HIP kernels (csrc/*.hip) through
the C ABI of include/ssg_hip.h; the only N x N objects that live in HBM are the half
matrices D (original distance) and J' (scaled Jaccard), see DESIGN.md.

`re_ranking_device` is the fused path used by `compute_dist` -> `generate_selflabel`: it
returns a `DistHandle` that stays on the GPU (no f64 N x N materialisation, no D2H).
"""
import numpy as np
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

_PAD = 32  # feature dim is zero-padded to a multiple of the Gram kernel's K step


class ReRankNaNError(ValueError):
    """max(source_dist_vec) == 0: the reference divides 0/0 at reid/rerank.py:40 and silently
    returns an all-NaN final_dist (sklearn then rejects the NaN eps).  Raised explicitly here."""


class DistHandle:
    """Device-resident result of one re_ranking call (one feature split).

    mode 0: final_dist[i,k] = f64(Jp[i,k]) + f64(half(v[i]+v[k])) * lambda   (rerank.py:122)
    mode 1: no-rerank: the half euclidean matrix itself (rerank.py:65-66)
    mode 2: an arbitrary float64 matrix uploaded by the caller (sklearn drop-in case)
    Rows [row0, row0+nrows) of the N x N problem are held locally (row-block sharding).

    Sharded handles (group is not None): `validate`, `final_dist`, `cluster.eps_rule`, `cluster.eps_rule_dbscan` and `DBSCAN.fit` are
    COLLECTIVE calls -- the status words of every rank are gathered so that all ranks take the same decision -- and must be made by
    every rank of the group, in the same order (a handle created with validate=False and materialised on one rank only would wait
    for its peers forever).
    """

    def __init__(self, N, mode, M, v=None, lambda_value=0.0, euclid=None, row0=0, nrows=None, group=None):
        self.N, self.mode, self.M, self.v, self.lambda_value = int(N), int(mode), M, v, float(lambda_value)
        self.euclid = euclid
        self.row0, self.nrows = int(row0), int(N if nrows is None else nrows)
        self.group = group   # torch.distributed group when the rows are sharded over ranks

    @property
    def device(self):
        return self.M.device

    _pending = None
    _error = None
    _redo = None
    _k1 = None
    sparse = None          # the sparse copy S of J' (re_ranking_device), or None; its consumers gate on the device-side overflow word

    def validate(self):
        """Raise what the pipeline could only detect on the device.  The status words are read once (with the caller's host
        round trip); a failure is kept on the handle and raised again by every later call, so a caller that catches the
        error cannot go on to cluster the NaN matrix.  Sharded rows: the words of every rank are gathered first (one small
        collective), so that all ranks take the same decision (a redo of the query expansion runs collectives)."""
        if self._pending is not None:
            self.resolve_pending(self.global_status(self.take_pending()))
        if self._error is not None:
            raise self._error
        return self

    def global_status(self, words, table=None, first=0):
        """status words as python numbers with the cross-rank ones replaced by their maximum over the ranks: digit overflow, "a V row was
        longer than the guess" and the longest V row must lead every rank to the same decision.  `words`: this rank's device tensor (read
        here, sharded: after ONE gather of every rank's words) -- or `table`: rows already gathered and read by the caller (cluster.eps_rule
        appends the words to its own status row; `first` = their offset in a row)."""
        if table is None:
            if self.group is None:
                return words.tolist()
            from .dist import gather_rows
            table, first = gather_rows(words.view(1, -1), self.group).tolist(), 0
        import torch.distributed as dist
        vals = [int(x) for x in table[dist.get_rank(self.group)][first:]]
        for j in (1, 2, 3):
            if len(vals) > j:
                vals[j] = max(int(r[first + j]) for r in table)
        return vals

    def take_pending(self):
        """the device status words as ONE int64 tensor (or None when they were read already): a consumer that is about to read
        something else back appends them to ITS read (cluster.eps_rule does) and hands the values to resolve_pending -- one host
        round trip less.  Words: max(v) half bits, digit overflow, V row longer than the guess, longest V row, S entries, S overflow."""
        if self._pending is None:
            return None
        words = [self._pending.to(torch.int64)]
        if self.sparse is not None:
            words.append(self.sparse["cursor"])
        return torch.cat(words) if len(words) > 1 else words[0]

    def pending_pieces(self):
        """the same words as `take_pending` as the RAW device tensors ([status int32 x 4] and, with a sparse copy, [S cursor int64 x 2]) --
        a consumer that copies its results out with DMA transfers adds these pieces to them instead of paying a concatenation launch;
        the values go to `resolve_pending` in the same order.  None when they were read already."""
        if self._pending is None:
            return None
        return [self._pending] + ([self.sparse["cursor"]] if self.sparse is not None else [])

    def resolve_pending(self, values):
        """values = the status words as python numbers (read with the caller's host round trip).  Returns True when the distance
        matrix was REBUILT (a V row was longer than the guessed capacity of the query expansion): kernels the caller queued on the
        old contents before reading the words must be run again."""
        redone = False
        if self._pending is not None:
            vmax_h, flag_h = int(values[0]), int(values[1])
            over, seen = (int(values[2]), int(values[3])) if len(values) >= 4 else (0, 0)
            self._pending = None
            if seen > 0 and self._k1 is not None:
                # (sharded rows: `seen` is the maximum over the ranks -- `global_status` -- so every rank stores the same guess)
                _QE_GUESS[_qe_key(self._k1, self.group)] = max(32, ((seen * 5 // 4) + 7) // 8 * 8)       # + 25 %: the LDS staging (and the occupancy) follows the guess
            if over and self._redo is not None:
                self._redo(seen)                     # (the sparse copy is rebuilt as well; its consumers gate on its device-side overflow word)
                redone = True
            elif self.sparse is not None and len(values) >= 6 and int(values[5]) != 0:
                self.sparse = None                   # the pool overflowed: S is unusable -- free it, later consumers take the dense entry points directly
            self._redo = None
            if flag_h:
                self._error = _lib.SSGError("ssg_gram_i8_encode: a feature did not fit the digit count chosen from max|feat| (internal error)")
            elif int(vmax_h) & 0x7FFF == 0:
                self._error = ReRankNaNError("max(source_dist_vec) == 0: every target->source 1-exp(-d^2) rounds to 0 in float16; the "
                                             "reference (reid/rerank.py:40) would return an all-NaN final_dist")
        if self._error is not None:
            raise self._error
        return redone

    def sparse_complete(self):
        """True when the handle holds a complete sparse copy S (its pool did not overflow).  Reads the device word: a blocking read,
        for tests and diagnostics -- the product's consumers gate on that word on the device."""
        return self.sparse is not None and int(self.sparse["cursor"][1].item()) == 0

    def final_dist(self):
        """float64 [nrows, N] device tensor (API materialisation, 8 bytes/entry)."""
        L = _lib.lib()
        self.validate()
        if self.mode == 2:
            return self.M
        if self.mode == 1:
            return self.M.to(torch.float64)
        out = torch.empty((self.nrows, self.N), dtype=torch.float64, device=self.device)
        check(L.ssg_final_dist_f64(ptr(self.M), ptr(self.v), self.N, self.row0, self.nrows, self.lambda_value, ptr(out), stream()),
              "ssg_final_dist_f64")
        return out


class DeviceBackedArray(np.ndarray):
    """numpy view of a materialised distance matrix that remembers its device handle, so
    `DBSCAN.fit_predict(final_dist)` / `generate_selflabel` can skip the re-upload.

    The handle is attached to the freshly materialised array ONLY (`attach`): views, copies, ufunc results and astype()
    never inherit it, and the array is handed out read-only, so every array that still carries a handle provably holds the
    device matrix's values.  Callers that want to edit the distances copy first (np.array(final)) or flip the flag back
    (final.setflags(write=True)); `cluster.as_handle` ignores the handle of a writeable array and uploads its contents."""
    ssg_handle = None

    def __array_finalize__(self, obj):
        self.ssg_handle = None

    @classmethod
    def attach(cls, arr, handle):
        out = np.asarray(arr).view(cls)
        out.ssg_handle = handle
        out.setflags(write=False)
        return out

    def valid_handle(self):
        return self.ssg_handle if (self.ssg_handle is not None and not self.flags.writeable) else None


def _as_dev_f32(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    x = x.to(device=device, dtype=torch.float32).contiguous()
    d = x.shape[1]
    if d % _PAD:
        x = torch.nn.functional.pad(x, (0, _PAD - d % _PAD))   # zeros add +0.0 exactly to every distance
    return x


def range_stats(tgt, src=None):
    """[max|tgt|, max|src|, max row norm of tgt, max row norm of src] as python floats: ONE kernel + ONE blocking read (the norms are
    float32 upper bounds).  Every host-side choice of the pipeline follows from them: digit count of the exact Gram, operand scales
    and the rigorous tolerance of the source term's bound pass."""
    L = _lib.lib()
    rs = torch.empty(4, dtype=torch.float32, device=tgt.device)
    check(L.ssg_range_stats_f32(ptr(tgt), tgt.shape[0], None if src is None else ptr(src), 0 if src is None else src.shape[0], tgt.shape[1], ptr(rs),
                                stream()), "ssg_range_stats_f32")
    return rs.tolist()


def source_vector(src, tgt, row0=0, nrows=None, exact_gemm=None, stats=None):
    """reid/rerank.py:35-40 on device -> (rowmin uint32-as-int32 [nrows]) for a row block.

    Default: filter-and-refine (a matrix-core pass bounds every distance per 4- or 8-source granule -- split-half
    operands on the fp16 cores, or float32 MFMA with SSG_SOURCE_BOUND=f32 -- then float64 re-evaluates the
    granules that can still hold the minimum): the exact minimum of the half-rounded float64 distances
    at a fraction of the cost of the full float64 Gram.  exact_gemm=True (or SSG_SOURCE_EXACT_GEMM=1)
    forces the full fp64-MFMA pass.  stats: optional (max|tgt|, max|src|, max target row norm, max source row norm)
    upper bounds already on the host (saves the device round trip)."""
    import os
    L = _lib.lib()
    N, d = tgt.shape
    nrows = N if nrows is None else nrows
    if exact_gemm is None:
        exact_gemm = os.environ.get("SSG_SOURCE_EXACT_GEMM", "0") == "1"
    rowmin = torch.empty(nrows, dtype=torch.int32, device=tgt.device)
    tblk = tgt[row0:row0 + nrows]
    Ns = src.shape[0]
    if not exact_gemm and (nrows * d * 4) < 2 ** 31:
        npad = (-Ns) % 256
        srcp = torch.nn.functional.pad(src, (0, 0, 0, npad)) if npad else src
        # rigorous float32 error bound of |x|^2 + |y|^2 - 2<x,y>: the MFMA dot is a d-term fmaf chain
        # (|err| <= gamma_d * sum|x_k y_k| <= gamma_d |x||y|, gamma_d = d*u/(1-d*u), u = 2^-24); the squared
        # norms come from a 32-term chain + 6-level tree (38 u relative); a few ulps for the final adds.
        if stats is None:      # (max|tgt|, max|src|, max row norm of the block, max row norm of src): one host read
            stats = range_stats(tblk.contiguous(), src)
        mt, ms, nx, ny = (float(s) for s in stats)
        u = 2.0 ** -24
        bound = os.environ.get("SSG_SOURCE_BOUND", "half")
        split = bound in ("split", "half")
        one = bound == "half"        # bound pass on the hi halves only (plain fp16 GEMM); "split" = three products (fp32-class)
        st = ss = 0.0
        if split:
            # bound pass on the fp16 matrix cores (split-half operands): per product 3*2^-22 relative (two operand
            # representations + the dropped lo*lo term) + 2^-24 absolute per operand below the half normals; the
            # accumulation is at most a 3d-term chain (three MFMA products per term), whatever order the hardware adds in
            import math
            st = 256.0 if mt == 0 else min(256.0, 2.0 ** math.floor(math.log2(16384.0 / mt)))
            ss = 256.0 if ms == 0 else min(256.0, 2.0 ** math.floor(math.log2(16384.0 / ms)))
            gam = 3.0 * d * u * 1.01 / (1.0 - 3.0 * d * u)
            dot_err = (gam + 12.0 * u) * nx * ny + u * math.sqrt(d) * (nx / ss + ny / st) * 1.01
            if one:
                # hi halves only: each operand carries a relative error 2^-11 (+ 2^-25 absolute below the half normals, covered
                # by the second term above), so |x.y - xh.yh| <= (2^-10 + 2^-22) sum|x_k y_k| <= 2^-10 * 1.001 |x||y|; the
                # accumulation is a d-term float32 chain (covered by gam)
                dot_err += (2.0 ** -10) * 1.001 * nx * ny
        else:
            gam = d * u / (1.0 - d * u)
            dot_err = gam * nx * ny
        tol = 2.0 * (2.0 * dot_err + 40.0 * u * (nx * nx + ny * ny)) + 8.0 * u * (nx + ny) ** 2
        nws = nrows + Ns + npad + nrows * ((Ns + npad) // 8) + ((nrows + Ns + npad) * d if split else 0)
        ws = torch.empty(nws, dtype=torch.float32, device=tgt.device)
        fn = L.ssg_source_rowmin_filtered1 if (split and one) else L.ssg_source_rowmin_filtered
        check(fn(ptr(tblk), ptr(srcp), nrows, Ns, Ns + npad, d, tol, st, ss, ptr(ws), ptr(rowmin), stream()), "ssg_source_rowmin_filtered")
        return rowmin
    ntgt = torch.empty(N, dtype=torch.float64, device=tgt.device)
    nsrc = torch.empty(Ns, dtype=torch.float64, device=tgt.device)
    check(L.ssg_row_norms_f64(ptr(tgt), N, d, 0, ptr(ntgt), stream()), "ssg_row_norms_f64")
    check(L.ssg_row_norms_f64(ptr(src), Ns, d, 0, ptr(nsrc), stream()), "ssg_row_norms_f64")
    check(L.ssg_source_rowmin_f16(ptr(tblk), ptr(ntgt[row0:]), ptr(src), ptr(nsrc), nrows, Ns, d, ptr(rowmin), stream()),
          "ssg_source_rowmin_f16")
    return rowmin


def _original_distance(L, tgt, row0, nrows, max_abs, st, memory_save=False, flag=None):
    """rows [row0,row0+nrows) of the half original distance (rerank.py:33,61-62) + their maxima -> (D, rowmax, flag).
    max_abs = max|tgt| on the host; flag = device flag of the int8 encoder (None on the fp64 path), to be read with the
    caller's next host round trip (it cannot be set after the range check here)."""
    dev = tgt.device
    N, d = tgt.shape
    D = torch.empty((nrows, N), dtype=torch.float16, device=dev)
    rowmax = torch.empty(nrows, dtype=torch.int32, device=dev)
    use_i8 = os.environ.get("SSG_SELF_GRAM", "i8") == "i8" and d <= 16384
    if use_i8:
        # exact integer Gram on the int8 matrix cores (half-rounded features in [-1, 1]: scipy's float64 sum is exact);
        # 3 radix-256 digits cover |feat| <= 0.498 (any real L2-normalised embedding), 4 digits |feat| <= 1
        mx = float(max_abs)
        nd = int(os.environ.get("SSG_SELF_GRAM_DIGITS", "0")) or (3 if mx <= 0.49 else 4)
        use_i8 = mx <= 1.0                       # (False for NaN as well)
    if use_i8:
        enc = torch.empty(L.ssg_gram_i8_encoded_bytes(N, d, nd), dtype=torch.int8, device=dev)
        inorm = torch.empty(N, dtype=torch.int64, device=dev)
        if flag is None:
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
        check(L.ssg_gram_i8_encode(ptr(tgt), N, d, nd, ptr(enc), ptr(inorm), ptr(flag), st), "ssg_gram_i8_encode")
        check(L.ssg_sqdist_self_i8(ptr(enc), ptr(inorm), N, d, nd, row0, nrows, int(bool(memory_save)), ptr(D), ptr(rowmax), ptr(flag), st), "ssg_sqdist_self_i8")
        del enc, inorm
    else:
        norms = torch.empty(N, dtype=torch.float64, device=dev)
        check(L.ssg_row_norms_f64(ptr(tgt), N, d, 1, ptr(norms), st), "ssg_row_norms_f64")
        check(L.ssg_sqdist_self_f16(ptr(tgt), ptr(norms), N, d, row0, nrows, int(bool(memory_save)), ptr(D), ptr(rowmax), st), "ssg_sqdist_self_f16")
        flag = None
    return D, rowmax, flag


from .dist import gather_rows as _gather_rows, gather_rows_packed as _gather_rows_packed  # noqa: E402,F401


RANK_MODES = ("introsort", "stable")


def default_rank_mode():
    """'introsort' = the unmodified reference's np.argsort tie order (reid/rerank.py:70, numpy's unstable introsort replayed on
    the device); 'stable' = canonical (value, column) order (np.argsort(kind='stable')), cheaper, opt-in (SSG_RANK_MODE=stable)."""
    mode = os.environ.get("SSG_RANK_MODE", "introsort")
    if mode not in RANK_MODES:
        raise ValueError("SSG_RANK_MODE must be one of %r" % (RANK_MODES,))
    return mode


def initial_rank(D, rowmax, N, nrows, K, rank_mode=None, force_arena=False, diag=None):
    """rerank.py:68-70 for a row block: int32 [nrows, K] = argsort(half(D / rowmax))[:, :K] in the requested tie order.
    force_arena (tests): run the introsort kernel from its global-memory arena even when a row fits in LDS.
    diag (tests): a dict that receives 'flagged' = int32 [nrows] device tensor, 1 for the rows the streamed replay handed to the
    in-place kernel (None when the streamed kernel did not run)."""
    L = _lib.lib()
    rank_mode = default_rank_mode() if rank_mode is None else rank_mode
    rank_blk = torch.empty((nrows, K), dtype=torch.int32, device=D.device)
    if rank_mode == "stable":
        check(L.ssg_topk_rank(ptr(D), ptr(rowmax), N, nrows, K, ptr(rank_blk), stream()), "ssg_topk_rank")
    elif rank_mode == "introsort":
        nws = int(L.ssg_topk_rank_introsort_arena_bytes(N, nrows) if force_arena else L.ssg_topk_rank_introsort_ws_bytes(N, nrows))
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=D.device)
        check(L.ssg_topk_rank_introsort(ptr(D), ptr(rowmax), N, nrows, K, ptr(rank_blk), ptr(ws), nws, stream()), "ssg_topk_rank_introsort")
        if diag is not None:
            off = int(L.ssg_topk_rank_introsort_flags_offset(N, nrows))
            full = int(L.ssg_topk_rank_introsort_arena_bytes(N, nrows))
            diag["flagged"] = ws[off:off + 4 * nrows].view(torch.int32).clone() if (off != 2 ** 64 - 1 and nws >= full) else None
    else:
        raise ValueError("rank_mode must be one of %r" % (RANK_MODES,))
    return rank_blk


_SIDE_STREAMS = {}


def _side_stream(dev):
    """the second HIP stream of a device (created once): the source term of `re_ranking_device` runs on it beside the k-reciprocal kernels"""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.2, no_rerank=False, keep_euclid=True, row0=0, nrows=None,
                      group=None, stages=None, rank_mode=None, memory_save=False, validate=True):
    """Fused device pipeline K3..K9 for one feature split.

    src [Ns,d], tgt [N,d]: float32 CUDA tensors (replicated on every rank of `group`).
    With a group, each rank computes rows [row0,row0+nrows) = its `dist.shard_bounds` block (N need not divide by the
    world size) and the small tables (rank lists, sparse V / V_qe, v) are all-gathered.
    `stages` (dict) receives intermediate tensors for parity tests.
    """
    L = _lib.lib()
    dev = tgt.device
    N = tgt.shape[0]
    nrows = N if nrows is None else nrows
    if N < 2:
        raise ValueError("re_ranking needs at least 2 target samples")
    if not no_rerank and (k1 < 1 or k2 < 1 or max(k1 + 1, k2) > 64):
        raise ValueError("re_ranking on the GPU supports 1 <= k1 <= 63 and 1 <= k2 <= 64 (got k1=%r, k2=%r)" % (k1, k2))
    src = _as_dev_f32(src, dev); tgt = _as_dev_f32(tgt, dev)
    d = tgt.shape[1]
    st = stream()

    # every range / norm bound the pipeline needs from the host: ONE kernel (ssg_range_stats_f32), ONE device round trip
    stats = range_stats(tgt, None if no_rerank else src)     # host round trip 1: max|tgt|, max|src|, max row norms (upper bounds)
    # ---- original distance (rerank.py:33,61-62): D half [nrows,N] + row max
    status = torch.zeros(4, dtype=torch.int32, device=dev)    # [max(v) half bits, digit overflow, V row longer than the guess, longest V row]
    D, rowmax, flag = _original_distance(L, tgt, row0, nrows, stats[0], st, memory_save, flag=status[1:2])
    if no_rerank:
        if flag is not None and int(flag.item()):
            raise _lib.SSGError("ssg_gram_i8_encode: a feature did not fit the digit count chosen from max|feat| (internal error)")
        return DistHandle(N, 1, D, euclid=D, row0=row0, nrows=nrows, group=group)

    # ---- source-domain term (rerank.py:35-40): v half [N]; initial ranking (rerank.py:68-70): the columns that are ever read are
    # [0, max(k1+1, k2)) (:76, :83, :97).  Sharded rows: the two row-block tables (source minima, rank lists) travel in ONE collective
    K = min(max(k1 + 1, k2), N)
    v = torch.empty(N, dtype=torch.float16, device=dev)
    vmax = status[0:1]
    # The source term (two matrix-core passes, ~1.3 ms at the bench's shape) is needed by nothing before the eps rule, and the kernels
    # between the initial ranking and the Jaccard rows (k-reciprocal sets, query expansion, inverted index: latency-bound, little LDS)
    # leave the matrix pipes idle -- it runs on a second stream behind the initial ranking.  One GPU: joined after the Jaccard rows.
    # Sharded rows (round 6): the block's minima travel with the LAST table gather in front of the Jaccard rows (V_qe, or V when k2 == 1)
    # instead of with the rank lists -- the same number of collectives, and the source term overlaps the k-reciprocal kernels, the V gather
    # and the query expansion.  SSG_RERANK_OVERLAP=0: everything on one stream, the minima with the rank lists (round 5's order).
    overlap = os.environ.get("SSG_RERANK_OVERLAP", "1") != "0"
    rank = initial_rank(D, rowmax, N, nrows, K, rank_mode)
    main = side = None
    src_state = {"rowmin": None, "joined": False, "gathered": group is None}
    if overlap:
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            src_state["rowmin"] = source_vector(src, tgt, row0, nrows, stats=stats)
    else:
        src_state["rowmin"] = source_vector(src, tgt, row0, nrows, stats=stats)

    def join_side():
        """main waits for the source term (idempotent); its result becomes usable on main"""
        if overlap and not src_state["joined"]:
            main.wait_stream(side)
            src_state["rowmin"].record_stream(main)
        src_state["joined"] = True

    def finish_source():
        join_side()
        check(L.ssg_source_vec_finish(ptr(src_state["rowmin"]), N, ptr(v), ptr(vmax), st), "ssg_source_vec_finish")

    def gather_with_source(tables):
        """the row-sharded `tables` in one collective -- together with the source minima of the block when they have not travelled yet"""
        if src_state["gathered"]:
            return _gather_rows_packed(tables, group, N)
        join_side()
        out = _gather_rows_packed([src_state["rowmin"]] + list(tables), group, N)
        src_state["rowmin"], src_state["gathered"] = out[0], True
        return out[1:]

    try:
        if group is not None:
            if overlap:
                rank, = _gather_rows_packed([rank], group, N)
            else:
                rank, = gather_with_source([rank])
                finish_source()

        # ---- k-reciprocal encoding (rerank.py:74-92)
        capV = int(L.ssg_krecip_row_capacity(k1))
        v_idx = torch.empty((nrows, capV), dtype=torch.int32, device=dev)
        v_val = torch.empty((nrows, capV), dtype=torch.float16, device=dev)
        v_nnz = torch.empty(nrows, dtype=torch.int32, device=dev)
        check(L.ssg_krecip(ptr(D), ptr(rowmax), ptr(rank), N, row0, nrows, K, k1, capV, ptr(v_idx), ptr(v_val), ptr(v_nnz), st), "ssg_krecip")
        if k2 == 1:
            v_idx, v_val, v_nnz = gather_with_source([v_idx, v_val, v_nnz])           # (the last gather in front of the Jaccard rows)
        else:
            v_idx, v_val, v_nnz = _gather_rows_packed([v_idx, v_val, v_nnz], group, N)       # (index / value / length rows: one collective)
        om = L.ssg_double_to_half_bits(1.0 - float(lambda_value))
        Jp = torch.empty((nrows, N), dtype=torch.float16, device=dev)
        tables = {}
        # sparse copy S of J' (the columns a row's Jaccard walk touches; everything else is the constant J'(0) = half(1 - lambda)): what the
        # eps rule and the region query walk instead of the N x N matrix while their bound stays below J'(0)
        sparse = None
        # (never usable for lambda > 1 -- J' would be negative, bit 15 is the kernel's "touched" marker -- or with the first-generation Jaccard kernel)
        if os.environ.get("SSG_SPARSE", "1") != "0" and not (om & 0x8000) and os.environ.get("SSG_JACCARD_GEN", "2") == "2":
            nseg = int(L.ssg_jaccard_segments(N))
            s_cap = int(nrows) * int(min(N, int(os.environ.get("SSG_SPARSE_ROW_ENTRIES", "1024"))))
            sparse = dict(pool=torch.empty(max(s_cap, 1), dtype=torch.int32, device=dev), cap=s_cap, cursor=torch.zeros(2, dtype=torch.int64, device=dev),
                          seg_off=torch.empty(nrows * nseg, dtype=torch.int64, device=dev), seg_len=torch.empty(nrows * nseg, dtype=torch.int32, device=dev),
                          nseg=nseg, jp0=int(om), vmin=torch.empty(1, dtype=torch.int32, device=dev), rowmask=torch.empty(nrows, dtype=torch.uint8, device=dev))

        def tail(mx, over):
            """local query expansion -> inverted index -> Jaccard rows into Jp, with LDS / row capacities sized for V rows of at most `mx`
            entries; `over` = the device words that report a longer row (None: mx is exact)"""
            if k2 != 1:
                kk = min(k2, N, K)
                capQ = kk * mx
                q_idx = torch.empty((nrows, capQ), dtype=torch.int32, device=dev)
                q_val = torch.empty((nrows, capQ), dtype=torch.float16, device=dev)
                q_nnz = torch.empty(nrows, dtype=torch.int32, device=dev)
                check(L.ssg_query_expand(ptr(v_idx), ptr(v_val), ptr(v_nnz), ptr(rank), N, row0, nrows, K, k2, capV, capQ, mx, ptr(q_idx), ptr(q_val),
                                         ptr(q_nnz), ptr(over), st), "ssg_query_expand")
                # sharded rows: the fixed-capacity rows travel as they are (k2 * longest V row entries of 6 bytes; trimming them to the longest
                # V_qe row would cost an all-reduce and a blocking read per split for a few MB over xGMI)
                q_idx, q_val, q_nnz = gather_with_source([q_idx, q_val, q_nnz])
            else:
                capQ, q_idx, q_val, q_nnz = capV, v_idx, v_val, v_nnz
            # ---- inverted index + Jaccard rows (rerank.py:101-122)
            # the inverted lists hold at most one entry per stored V_qe entry: sized by that bound instead of reading sum(q_nnz) back
            total = int(N) * int(capQ)
            colcnt = torch.empty(N, dtype=torch.int32, device=dev)
            colptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
            inv_row = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            inv_val = torch.empty(max(total, 1), dtype=torch.float16, device=dev)
            check(L.ssg_invert_index(ptr(q_idx), ptr(q_val), ptr(q_nnz), N, N, capQ, ptr(colcnt), ptr(colptr), ptr(inv_row), ptr(inv_val), st),
                  "ssg_invert_index")
            colmeta = torch.empty((2, nrows, capQ), dtype=torch.int32, device=dev)
            if sparse is not None:
                check(L.ssg_jaccard_rows2(ptr(q_idx), ptr(q_val), ptr(q_nnz), capQ, ptr(colptr), ptr(inv_row), ptr(inv_val), total, ptr(colmeta), N, row0, nrows,
                                          om, ptr(Jp), ptr(sparse["pool"]), sparse["cap"], ptr(sparse["cursor"]), ptr(sparse["seg_off"]), ptr(sparse["seg_len"]), st),
                      "ssg_jaccard_rows2")
            else:
                check(L.ssg_jaccard_rows(ptr(q_idx), ptr(q_val), ptr(q_nnz), capQ, ptr(colptr), ptr(inv_row), ptr(inv_val), total, ptr(colmeta), N, row0, nrows,
                                         om, ptr(Jp), st), "ssg_jaccard_rows")
            if stages is not None:
                tables.update(q_idx=q_idx, q_val=q_val, q_nnz=q_nnz, colptr=colptr, inv_row=inv_row, inv_val=inv_val)

        # ---- local query expansion (rerank.py:94-99) .. Jaccard rows.  The LDS staging and the row capacity of V_qe follow the longest V
        # row.  Run on a GUESS (the longest row of the previous call + 25 %, 64 at first) and let the kernel report a longer row
        # through the status words that the consumer reads anyway (`validate`); a miss redoes this tail with the exact bound (rare; the
        # result is the same either way).
        redo = None
        if k2 == 1:
            tail(capV, None)
        elif os.environ.get("SSG_QE_GUESS", "1") != "0":
            # (sharded rows as well since round 5: every rank runs on the same guess -- kept per group, `_qe_key` -- and the words that report
            # a miss are combined over the ranks -- maximum -- when they are read, `DistHandle.global_status`, so that all ranks redo
            # together; round 4 read the gathered v_nnz back here instead: one blocking read per split)
            guess = min(capV, _QE_GUESS.get(_qe_key(k1, group), 64))
            tail(guess, status[2:4])
            # a miss redoes the tail sized by the longest row the kernel REPORTED (status[3]: exact over the rows it staged), not by the worst-case
            # capacity capV = (k1+1)(round(k1/2)+2) -- that one passes the 160 KB of LDS from k1 ~ 35 on; 0 (no report) falls back to a read of v_nnz
            redo = lambda seen=0: tail(max(int(seen), 1) if seen else max(int(v_nnz.max().item()), 1), None)      # noqa: E731  (keeps the small tables alive, not D)
        else:
            tail(max(int(v_nnz.max().item()), 1), None)  # exact bound: one blocking read (v_nnz is the gathered table: the same on every rank)

        if not (group is not None and not overlap):
            finish_source()                              # (one stream + sharded rows: v was finished with the rank lists)
        if sparse is not None:
            check(L.ssg_half_min(ptr(v), N, ptr(sparse["vmin"]), st), "ssg_half_min")      # the row floors J'(0) + lambda * half(v_i + min v) of the sparse passes
    finally:
        # whatever happened above (e.g. the query expansion's LDS limit at a large k1): main never runs ahead of the side stream, so that
        # the caller cannot free src / tgt while the source term still reads them
        if overlap and not src_state["joined"]:
            main.wait_stream(side)
    h = DistHandle(N, 0, Jp, v=v, lambda_value=lambda_value, euclid=D if keep_euclid else None, row0=row0, nrows=nrows, group=group)
    # the device-side status words (zero source vector -> the reference's NaN path; int8 digit overflow; a V row longer than the guess)
    # are read with the consumer's first host round trip (`validate`: eps_rule / DBSCAN / final_dist), not with one of their own
    h._pending, h._redo, h._k1 = status, redo, k1
    h.sparse = sparse
    if stages is not None and sparse is not None:
        tables["sparse"] = sparse
    if validate or stages is not None:
        h.validate()
    if stages is not None:
        stages.update(D=D, rowmax=rowmax, v=v, rank=rank, v_idx=v_idx, v_val=v_val, v_nnz=v_nnz, Jp=Jp, **tables)
    return h


# (k1, members of the group or None) -> guessed longest V row for the next call (the longest row of the last call + 25 %, a multiple
# of 8).  The guess sizes tables that travel through a flat all-gather, so every rank of a group must hold the SAME number: the sharded
# entries are written only from status words combined over the ranks (`DistHandle.global_status`) and never shared with the entries of
# one-GPU calls, whose `seen` is rank-local (a rank that had run an extra one-GPU re-rank would have sent tables of another
# width into the gather).
_QE_GUESS = {}


def _qe_key(k1, group):
    if group is None:
        return (int(k1), None)
    import torch.distributed as dist
    return (int(k1), tuple(dist.get_process_group_ranks(group)))


def re_ranking(input_feature_source, input_feature, k1=20, k2=6, lambda_value=0.2, MemorySave=False, Minibatch=2000, no_rerank=False,
               device=None, rank_mode=None):
    """Drop-in for reid/rerank.py:27 re_ranking (numpy in, numpy out).

    MemorySave=True selects the numerics of the reference's chunked branch (rerank.py:49-59: the original distance is
    squared in float64 and rounded to half once); Minibatch only sizes the reference's row chunks and changes no value, so
    it is accepted and ignored (no chunking is needed on a 288 GB device).  Tie order of the initial ranking is the reference's own
    (np.argsort default = numpy's unstable introsort, replayed on the device); rank_mode='stable'
    selects the canonical (value, index) order instead, see DESIGN.md "ties".
    """
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    print('computing source distance...')
    print('computing original distance...')
    src = _as_dev_f32(np.asarray(input_feature_source), device)
    tgt = _as_dev_f32(np.asarray(input_feature), device)
    if not no_rerank:
        print('starting re_ranking...')
    h = re_ranking_device(src, tgt, k1=k1, k2=k2, lambda_value=lambda_value, no_rerank=no_rerank, rank_mode=rank_mode,
                          memory_save=MemorySave)
    # the literal numpy return (rerank.py:127: 2 N^2 + 8 N^2 bytes) at PCIe speed: page-locked destinations from a pool, the float64
    # matrix produced chunk by chunk while the previous chunk travels (hostio.py); the arrays own their memory like the reference's
    from . import hostio
    if no_rerank:
        return hostio.to_host(h.euclid).numpy(), None
    h.validate()
    euclid = hostio.to_host(h.euclid, wait=False)
    final = DeviceBackedArray.attach(hostio.final_dist_to_host(h).numpy(), h)      # (waits for the copy stream: both copies have landed)
    return euclid.numpy(), final


def _init_pipeline(D, nq, k1, k2, lambda_value):
    """sparse stages of re_ranking_init on a float32 distance matrix D [N,N] (device)."""
    L = _lib.lib()
    dev, st = D.device, stream()
    N = D.shape[0]
    K = min(k1 + 1, N)
    capV = int(L.ssg_krecip_row_capacity(k1))
    rowmax = torch.empty(N, dtype=torch.float32, device=dev)
    rank = torch.empty((N, K), dtype=torch.int32, device=dev)
    v_idx = torch.empty((N, capV), dtype=torch.int32, device=dev); v_val = torch.empty((N, capV), dtype=torch.float32, device=dev)
    v_nnz = torch.empty(N, dtype=torch.int32, device=dev)
    check(L.ssg_rerank_init_stage1(ptr(D), N, k1, k2, capV, ptr(rowmax), ptr(rank), ptr(v_idx), ptr(v_val), ptr(v_nnz), st), "ssg_rerank_init_stage1")
    if k2 != 1:
        kk = min(k2, N, K)
        mx = max(int(v_nnz.max().item()), 1)
        capQ = kk * mx
        q_idx = torch.empty((N, capQ), dtype=torch.int32, device=dev); q_val = torch.empty((N, capQ), dtype=torch.float32, device=dev)
        q_nnz = torch.empty(N, dtype=torch.int32, device=dev)
        check(L.ssg_rerank_init_expand(ptr(v_idx), ptr(v_val), ptr(v_nnz), ptr(rank), N, k1, k2, capV, capQ, mx, ptr(q_idx), ptr(q_val), ptr(q_nnz), st),
              "ssg_rerank_init_expand")
    else:
        capQ, q_idx, q_val, q_nnz = capV, v_idx, v_val, v_nnz
    total = max(int(q_nnz.sum().item()), 1)
    colcnt = torch.empty(N, dtype=torch.int32, device=dev); colptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
    inv_row = torch.empty(total, dtype=torch.int32, device=dev); inv_val = torch.empty(total, dtype=torch.float32, device=dev)
    out = torch.empty((nq, N - nq), dtype=torch.float32, device=dev)
    check(L.ssg_rerank_init_jaccard(ptr(D), ptr(rowmax), ptr(q_idx), ptr(q_val), ptr(q_nnz), capQ, N, nq, float(lambda_value), ptr(colcnt), ptr(colptr),
                                    ptr(inv_row), ptr(inv_val), ptr(out), st), "ssg_rerank_init_jaccard")
    return out


def re_ranking_init(query_feature, gallery_feature, k1=20, k2=6, lambda_value=0.3, device=None):
    """Drop-in for reid/rerank.py:171 re_ranking_init (float32 cosine variant; numpy in, numpy
    [num_query, num_gallery] float32 out).  The stacked Gram matrix 2 - 2 x.y runs on the fp32-MFMA GEMM."""
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    q = torch.as_tensor(np.asarray(query_feature, dtype=np.float32)); g = torch.as_tensor(np.asarray(gallery_feature, dtype=np.float32))
    nq, N = q.shape[0], q.shape[0] + g.shape[0]
    x = _as_dev_f32(torch.cat([q, g], 0), device)
    npad = (-N) % 64
    y = torch.nn.functional.pad(x, (0, 0, 0, npad)) if npad else x
    zeros = torch.zeros(N + npad, dtype=torch.float32, device=device)
    Dp = torch.empty((N, N + npad), dtype=torch.float32, device=device)
    check(L.ssg_cosine_dist_f32(ptr(x), ptr(y), N, N + npad, x.shape[1], ptr(zeros), ptr(Dp), stream()), "ssg_cosine_dist_f32")
    D = Dp[:, :N].contiguous() if npad else Dp
    return _init_pipeline(D, nq, k1, k2, lambda_value).cpu().numpy()


def re_ranking_init_dist(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3, device=None):
    """Drop-in for reid/rerank_initial.py:40 re_ranking_init (takes the dot-product matrices)."""
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    qg = torch.as_tensor(np.asarray(q_g_dist, dtype=np.float32)); qq = torch.as_tensor(np.asarray(q_q_dist, dtype=np.float32))
    gg = torch.as_tensor(np.asarray(g_g_dist, dtype=np.float32))
    dots = torch.cat([torch.cat([qq, qg], 1), torch.cat([qg.t(), gg], 1)], 0).to(device).contiguous()
    D = torch.empty_like(dots)
    check(L.ssg_affine_2m2x_f32(ptr(dots), ptr(D), dots.numel(), stream()), "ssg_affine_2m2x_f32")
    return _init_pipeline(D, qg.shape[0], k1, k2, lambda_value).cpu().numpy()
