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
    ok = ok and gr.shape == (3 * world, 4) and all(float(gr[3 * r:3 * r + 3].min()) == float(gr[3 * r:3 * r + 3].max()) == float(r) for r in range(world))
    # ragged row blocks (N not divisible by the world size): one flat all-gather of padded blocks, compacted in rank order
    for n in (11, 16522 % 1000, 7, world, world + 1):
        lo, hi = sd.shard_bounds(n, rank, world)
        table = torch.arange(n * 3, dtype=torch.int32).view(n, 3)
        ok = ok and torch.equal(sd.gather_rows(table[lo:hi].clone(), g, n), table)
        vec = torch.arange(n, dtype=torch.float16)
        ok = ok and torch.equal(sd.gather_rows(vec[lo:hi].clone(), g, n), vec)
    try:
        sd.gather_rows(torch.zeros((5, 2)), g, 11 if world == 2 and rank == 0 else 12)   # a block of the wrong size is refused
        ok = False
    except ValueError:
        pass
    h = torch.tensor([1, 2, 3], dtype=torch.int64) * (rank + 1)
    ok = ok and torch.equal(sd.all_reduce_sum(h, g), torch.tensor([1, 2, 3]) * (world * (world + 1) // 2))
    empty = sd.gather_varlen(torch.zeros((0, 2), dtype=torch.int32) if rank == 0 else torch.ones((2, 2), dtype=torch.int32), g)
    ok = ok and empty.shape == (2 * (world - 1), 2)
    sd.barrier(g)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_dist_plumbing_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(60) for p in procs]
    assert res == [(r, True) for r in range(world)]


def test_shard_bounds_cover():
    from ssg_amd.dist import shard_bounds
    for n in (1, 7, 16000, 12936):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _gpu_worker(rank, world, port, q, N, Ns, d):
    import torch.distributed as dist
    from types import SimpleNamespace
    import ssg_amd  # noqa: F401
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd import dist as sd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    Ns, d = Ns, d
    nsplit = 3 if N < 10000 else 1
    tgts = [torch.from_numpy(clustered(N, d, 5 + s)).to(dev) for s in range(nsplit)]
    srcs = [torch.from_numpy(clustered(Ns, d, 60 + s, intra=0.7)).to(dev) for s in range(nsplit)]
    lo, hi = sd.shard_bounds(N, rank, world)
    out = {}
    for mode, no_rerank in ((("rerank", False), ("norerank", True)) if N < 10000 else (("rerank", False),)):
        e_list, r_list = compute_dist(srcs, tgts, lambda_value=0.1, no_rerank=no_rerank, num_split=2, group=g)
        args = SimpleNamespace(no_rerank=no_rerank, rho=1.6e-3)
        labels, clusters = generate_selflabel(e_list, r_list, 0, args, [])
        hs = e_list if no_rerank else r_list
        assert all(h.row0 == lo and h.nrows == hi - lo for h in hs)
        import hashlib
        pack = (lambda m: hashlib.sha256(m.tobytes()).hexdigest()) if N > 10000 else (lambda m: m)
        out[mode] = [(float(c.eps), l, pack(h.M.cpu().numpy().view(np.uint16))) for c, l, h in zip(clusters, labels, hs)]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,N", [(2, 1536), (8, 1531), (8, 30003)])
def test_sharded_pipeline_matches_unsharded(world, N):
    """compute_dist -> generate_selflabel, 3 feature splits, rows sharded over `world` processes (gloo; they share the
    test box's single GPU) with ragged row blocks (1531 = 8*191 + 3; 30003 = BASELINE configs[3]'s MSMT17-size problem, one split
    over 8 ranks, with a non-divisible N): eps, labels and every local block of the distance matrix bit-identical to the
    unsharded run.  (The 8-GPU RCCL leg itself is the driver's; this is the same code over gloo.)"""
    from types import SimpleNamespace
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd.dist import shard_bounds
    Ns, d = (640, 96) if N < 10000 else (4000, 128)
    dev = torch.device("cuda", 0)
    nsplit = 3 if N < 10000 else 1           # the MSMT-size case: one split, re-rank mode (keeps the 8 processes on one GPU short)
    modes = (("rerank", False), ("norerank", True)) if N < 10000 else (("rerank", False),)
    tgts = [torch.from_numpy(clustered(N, d, 5 + s)).to(dev) for s in range(nsplit)]
    srcs = [torch.from_numpy(clustered(Ns, d, 60 + s, intra=0.7)).to(dev) for s in range(nsplit)]
    ref = {}
    for mode, no_rerank in modes:
        e_list, r_list = compute_dist(srcs, tgts, lambda_value=0.1, no_rerank=no_rerank, num_split=2)
        args = SimpleNamespace(no_rerank=no_rerank, rho=1.6e-3)
        labels, clusters = generate_selflabel(e_list, r_list, 0, args, [])
        hs = e_list if no_rerank else r_list
        ref[mode] = [(float(c.eps), l, h.M.cpu().numpy().view(np.uint16)) for c, l, h in zip(clusters, labels, hs)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, N, Ns, d)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=900) for _ in range(world))
    [p.join(120) for p in procs]
    for mode, _ in modes:
        for s in range(nsplit):
            e0, l0, m0 = ref[mode][s]
            for r in range(world):
                e, l, m = res[r][mode][s]
                lo, hi = shard_bounds(N, r, world)
                assert e == e0, (mode, s, r)
                assert np.array_equal(l, l0), (mode, s, r)
                if N > 10000:
                    import hashlib
                    assert m == hashlib.sha256(np.ascontiguousarray(m0[lo:hi]).tobytes()).hexdigest(), (mode, s, r)
                else:
                    assert np.array_equal(m, m0[lo:hi]), (mode, s, r)
