"""Multi-process tests of the row-block sharded path.

CPU (gloo, world_size 2): the collective plumbing of ssg_amd/dist.py and the shard arithmetic.
GPU (-m gpu; 2 processes sharing the single GPU of the test box over gloo): the sharded device
pipeline (re-rank -> eps -> DBSCAN) must reproduce the unsharded result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import clustered


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cpu_worker(rank, world, port, q):
    import torch.distributed as dist
    import ssg_amd  # noqa: F401
    from ssg_amd import dist as sd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    lo, hi = sd.shard_bounds(11, rank, world)
    rows = torch.arange(22, dtype=torch.int32).view(11, 2)[lo:hi]
    full = sd.gather_varlen(rows, g)
    ok = torch.equal(full, torch.arange(22, dtype=torch.int32).view(11, 2))
    blk = torch.full((3, 4), float(rank))
    gr = sd.gather_rows(blk, g)
    ok = ok and gr.shape == (6, 4) and float(gr[:3].max()) == 0.0 and float(gr[3:].min()) == 1.0
    h = torch.tensor([1, 2, 3], dtype=torch.int64) * (rank + 1)
    ok = ok and torch.equal(sd.all_reduce_sum(h, g), torch.tensor([3, 6, 9]))
    empty = sd.gather_varlen(torch.zeros((0, 2), dtype=torch.int32) if rank == 0 else torch.ones((2, 2), dtype=torch.int32), g)
    ok = ok and empty.shape == (2, 2)
    sd.barrier(g)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_dist_plumbing_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover():
    from ssg_amd.dist import shard_bounds
    for n in (1, 7, 16000, 12936):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _gpu_worker(rank, world, port, q, N, Ns, d):
    import torch.distributed as dist
    import ssg_amd  # noqa: F401
    from ssg_amd import cluster, rerank
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    tgt = torch.from_numpy(clustered(N, d, 5)).to(dev); src = torch.from_numpy(clustered(Ns, d, 6, intra=0.7)).to(dev)
    nrows = N // world
    out = {}
    for mode, kw in (("rerank", dict(lambda_value=0.1)), ("norerank", dict(no_rerank=True))):
        h = rerank.re_ranking_device(src, tgt, row0=rank * nrows, nrows=nrows, group=g, **kw)
        eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
        lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
        out[mode] = (float(eps), cnt, top, lab, h.M.cpu().numpy().view(np.uint16))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_pipeline_matches_unsharded():
    from ssg_amd import cluster, rerank
    N, Ns, d = 1536, 640, 96
    dev = torch.device("cuda", 0)
    tgt = torch.from_numpy(clustered(N, d, 5)).to(dev); src = torch.from_numpy(clustered(Ns, d, 6, intra=0.7)).to(dev)
    ref = {}
    for mode, kw in (("rerank", dict(lambda_value=0.1)), ("norerank", dict(no_rerank=True))):
        h = rerank.re_ranking_device(src, tgt, **kw)
        eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
        lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
        ref[mode] = (float(eps), cnt, top, lab, h.M.cpu().numpy().view(np.uint16))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, N, Ns, d)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=600) for _ in range(world))
    [p.join(120) for p in procs]
    for mode in ("rerank", "norerank"):
        e0, c0, t0, l0, m0 = ref[mode]
        for r in range(world):
            e, c, t, l, m = res[r][mode]
            assert (e, c, t) == (e0, c0, t0), (mode, r)
            assert np.array_equal(l, l0), (mode, r)
            assert np.array_equal(m, m0[r * (N // world):(r + 1) * (N // world)]), (mode, r)
