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
        # round 5: the same blocks of SEVERAL tables (different dtypes / row widths, incl. an odd byte count) in ONE collective
        idx = torch.arange(n * 5, dtype=torch.int32).view(n, 5); val = (torch.arange(n * 5, dtype=torch.float32).view(n, 5) / 7).to(torch.float16)
        nnz = torch.arange(n, dtype=torch.int32) + 3; by = (torch.arange(n * 3) % 251).to(torch.uint8).view(n, 3)
        got = sd.gather_rows_packed([idx[lo:hi].clone(), val[lo:hi].clone(), nnz[lo:hi].clone(), by[lo:hi].clone()], g, n)
        ok = ok and len(got) == 4 and all(torch.equal(a, b) and a.is_contiguous() for a, b in zip(got, (idx, val, nnz, by)))
    # equally shaped per-rank tables stacked by rank (the sharded DBSCAN ships count + neighbour counts + edge list like this)
    st = sd.gather_packed([torch.tensor([rank * 10 + 1], dtype=torch.int64), torch.full((4, 2), rank, dtype=torch.int32), torch.zeros((0, 2), dtype=torch.int32)], g)
    ok = ok and [int(x) for x in st[0].flatten()] == [r * 10 + 1 for r in range(world)] and st[1].shape == (world, 4, 2) and st[2].shape == (world, 0, 2)
    ok = ok and all(int(st[1][r].min()) == int(st[1][r].max()) == r for r in range(world))
    try:
        sd.gather_rows(torch.zeros((5, 2)), g, 11 if world == 2 and rank == 0 else 12)   # a block of the wrong size is refused
        ok = False
    except ValueError:
        pass
    h = torch.tensor([1, 2, 3], dtype=torch.int64) * (rank + 1)
    ok = ok and torch.equal(sd.all_reduce_sum(h, g), torch.tensor([1, 2, 3]) * (world * (world + 1) // 2))
    empty = sd.gather_varlen(torch.zeros((0, 2), dtype=torch.int32) if rank == 0 else torch.ones((2, 2), dtype=torch.int32), g)
    ok = ok and empty.shape == (2 * (world - 1), 2)
    if world >= 3:
        # two groups of different membership on one backend -- ranks 0, 1 gather over {0, 1} first, then everybody gathers
        # over the world group.  The flat-gather probe is a collective: it must run per group, or rank 2 probes alone and hangs.
        sub = dist.new_group(ranks=[0, 1])
        if rank < 2:
            part = sd.gather_rows(torch.full((2, 2), float(rank)), sub)
            ok = ok and part.shape == (4, 2) and float(part[2:].min()) == 1.0
        allr = sd.gather_rows(torch.full((1, 2), float(rank)), g)
        ok = ok and allr.shape == (world, 2) and [float(v) for v in allr[:, 0]] == [float(r) for r in range(world)]
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


def _gpu_worker(rank, world, port, q, N, Ns, d, grouping="shard"):
    import torch.distributed as dist
    from types import SimpleNamespace
    import ssg_amd  # noqa: F401
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd import dist as sd
    # (N == 1533, below) round 5: the sharded path runs the query expansion on a guessed row capacity too.  Force a miss (4 entries): the
    # words that report it are combined over the ranks when the eps rule reads them, every rank redoes the tail together (its collectives
    # stay matched), and the result equals the unsharded one
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    if N == 1533:
        from ssg_amd import rerank
        rerank._QE_GUESS[rerank._qe_key(20, g)] = 4
        # the guess of a group is kept apart from the one-GPU calls' (whose longest row is rank-local): a rank that
        # has run a one-GPU re-rank of its own must still size the gathered tables like its peers
        rerank._QE_GUESS[(20, None)] = 8 * (rank + 1)
    Ns, d = Ns, d
    nsplit = 3
    tgts = [torch.from_numpy(clustered(N, d, 5 + s)).to(dev) for s in range(nsplit)]
    srcs = [torch.from_numpy(clustered(Ns, d, 60 + s, intra=0.7)).to(dev) for s in range(nsplit)]
    lo, hi = sd.shard_bounds(N, rank, world)
    out = {}
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import CollectiveCounter, SyncCounter
    sd._flat_gather_supported(g)                 # (the one-off probe collective of the group, outside the count)
    for mode, no_rerank in ((("rerank", False), ("norerank", True)) if N < 10000 else (("rerank", False),)):
        with CollectiveCounter() as cc, SyncCounter() as sc:
            e_list, r_list = compute_dist(srcs, tgts, lambda_value=0.1, no_rerank=no_rerank, num_split=2, group=g, grouping=grouping)
            args = SimpleNamespace(no_rerank=no_rerank, rho=1.6e-3)
            labels, clusters = generate_selflabel(e_list, r_list, 0, args, [])
        ncoll = sum(v for k, v in cc.calls.items() if k != "barrier")
        if grouping == "replicate":
            assert ncoll == 0, cc.calls
        elif grouping == "shard" and mode == "rerank" and N in (1536, 1531, 30003):
            # VERDICT r5 next #5c: <= 6 collectives (3 table gathers of the re-rank + 2 of the eps rule / DBSCAN chain) and <= 3 blocking reads
            # per split when nothing has to be redone (N = 1533 forces a query-expansion miss and runs its tail twice)
            assert ncoll <= 6 * nsplit and sc.n <= 3 * nsplit, (cc.calls, sc.n)
        hs = e_list if no_rerank else r_list
        if grouping == "shard":
            assert all(h.row0 == lo and h.nrows == hi - lo and h.group is g for h in hs)
        elif grouping == "replicate" or sd.choose_grouping(N, world) == "replicate":
            assert all(h.row0 == 0 and h.nrows == N and h.group is None for h in hs)       # the whole problem on every rank, no collective
        import hashlib
        pack = (lambda m: hashlib.sha256(m.tobytes()).hexdigest()) if N > 10000 else (lambda m: m)
        out[mode] = [(float(c.eps), l, pack(h.M.cpu().numpy().view(np.uint16)[(lo - h.row0):(hi - h.row0)])) for c, l, h in zip(clusters, labels, hs)]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,N", [(2, 1536), (3, 1533), (8, 1531), (8, 30003)])
def test_sharded_pipeline_matches_unsharded(world, N):
    """compute_dist -> generate_selflabel, 3 feature splits, rows sharded over `world` processes (gloo; they share the
    test box's single GPU) with ragged row blocks (1531 = 8*191 + 3; 30003 = BASELINE configs[3]'s MSMT17-size problem, three
    splits over 8 ranks, with a non-divisible N): eps, labels and every local block of the distance matrix bit-identical to the
    unsharded run.  (The 8-GPU RCCL leg itself is the driver's; this is the same code over gloo.)"""
    from types import SimpleNamespace
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd.dist import shard_bounds
    Ns, d = (640, 96) if N < 10000 else (4000, 128)
    dev = torch.device("cuda", 0)
    nsplit = 3                               # whole / upper / lower feature sets (BASELINE configs[3]); the MSMT-size case runs the re-rank mode only
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


@pytest.mark.gpu
@pytest.mark.parametrize("grouping", ["replicate", "auto"])
def test_grouping_forms_give_the_sharded_labels(grouping):
    """VERDICT r5 next #5a: compute_dist(..., group=, grouping=) -- 'replicate' (every rank runs the whole problem, no collective),
    'shard' (row blocks + all-gathers) and 'auto' (dist.choose_grouping: N = 1536 on 2 ranks replicates) give the same eps, the
    same labels and the same distance rows on every rank as the one-GPU run."""
    from types import SimpleNamespace
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd.dist import shard_bounds
    world, N, Ns, d = 2, 1536, 640, 96
    dev = torch.device("cuda", 0)
    tgts = [torch.from_numpy(clustered(N, d, 5 + s)).to(dev) for s in range(3)]
    srcs = [torch.from_numpy(clustered(Ns, d, 60 + s, intra=0.7)).to(dev) for s in range(3)]
    ref = {}
    for mode, no_rerank in (("rerank", False), ("norerank", True)):
        e_list, r_list = compute_dist(srcs, tgts, lambda_value=0.1, no_rerank=no_rerank, num_split=2)
        labels, clusters = generate_selflabel(e_list, r_list, 0, SimpleNamespace(no_rerank=no_rerank, rho=1.6e-3), [])
        hs = e_list if no_rerank else r_list
        ref[mode] = [(float(c.eps), l, h.M.cpu().numpy().view(np.uint16)) for c, l, h in zip(clusters, labels, hs)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, N, Ns, d, grouping)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=900) for _ in range(world))
    [p.join(120) for p in procs]
    for mode in ref:
        for s in range(3):
            e0, l0, m0 = ref[mode][s]
            for r in range(world):
                e, l, m = res[r][mode][s]
                lo, hi = shard_bounds(N, r, world)
                assert e == e0 and np.array_equal(l, l0) and np.array_equal(m, m0[lo:hi]), (mode, s, r)


def test_grouping_policy_model():
    """dist.choose_grouping: a pure function of (N, world) -- every rank takes the same decision --, small problems replicate, large
    ones shard, a problem that does not fit one GPU shards whatever was asked, the environment overrides 'auto' only."""
    from ssg_amd import dist as sd
    assert sd.choose_grouping(16000, 1) == "replicate"
    assert sd.choose_grouping(2000, 8) == "replicate" and sd.choose_grouping(128000, 8) == "shard"
    assert sd.choose_grouping(16000, 8, "replicate") == "replicate" and sd.choose_grouping(2000, 8, "shard") == "shard"
    assert sd.choose_grouping(400000, 8, "replicate") == "shard"           # 5 N^2 bytes > 0.8 x 288 GB
    forms = [sd.choose_grouping(n, 8) for n in range(1000, 40000, 1000)]
    assert forms == sorted(forms)                                         # one crossover: 'replicate' ... then 'shard' ...
    t_rep, t_shard = sd.grouping_time_model(16000, 8)
    assert 4e-3 < t_rep < 8e-3 and t_shard < t_rep                        # the measured one-GPU leg is ~5.5 ms at N = 16 000
    os.environ["SSG_GROUPING"] = "replicate"
    try:
        assert sd.choose_grouping(128000, 8) == "replicate" and sd.choose_grouping(128000, 8, "shard") == "shard"
    finally:
        del os.environ["SSG_GROUPING"]
    with pytest.raises(ValueError):
        sd.choose_grouping(1000, 2, "both")


# ------------------------------------------------------------------ sharded feature extraction (images split by rank + C1 all-gather)
class _FakeEmbedder:
    """stands in for the HIP ResNet in the CPU plumbing test: a deterministic per-image function of the pixels"""
    num_split = 2
    device = torch.device("cpu")

    def eval(self):
        return self

    def embed_with_flip(self, x, for_eval=False):
        x = x.float()
        base = torch.stack([x.mean(dim=(1, 2, 3)), x.amax(dim=(1, 2, 3)), x[:, 0].sum(dim=(1, 2))], dim=1)      # [B, 3]
        feats = torch.stack([base.repeat(1, 683)[:, :2048] * (s + 1) for s in range(3)])                         # [3, B, 2048]
        return feats.permute(1, 0, 2).reshape(x.shape[0], -1).contiguous() if for_eval else feats


def _extract_cpu_worker(rank, world, port, q, n_img, batch):
    import torch.distributed as dist
    import ssg_amd  # noqa: F401
    from ssg_amd import evaluators as ev
    ev._check_model = lambda m: m
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    imgs = torch.randn(n_img, 3, 8, 4, generator=torch.Generator().manual_seed(3))
    names = ["img%05d.jpg" % i for i in range(n_img)]
    pids = [i % 7 for i in range(n_img)]
    model = _FakeEmbedder()
    ok = True
    for for_eval in (False, True):
        ref, rn, rp = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, batch, names, pids), for_eval=for_eval)
        got, gn, gp = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, batch, names, pids), for_eval=for_eval, group=g)
        ok = ok and torch.equal(ref, got) and rn == gn and rp == gp
        # a loader without shard(): the generic skip-foreign-batches path
        plain = list(ev.TensorBatchLoader(imgs, batch, names, pids))
        got2, gn2, _ = ev.extract_embeddings(model, plain, for_eval=for_eval, group=g)
        ok = ok and torch.equal(ref, got2) and rn == gn2
        loc, ln, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, batch, names, pids), for_eval=for_eval, group=g, gather=False)
        lo, hi = ev.TensorBatchLoader(imgs, batch).shard(rank, world).first, None
        ok = ok and ln == names[lo:lo + len(ln)] and loc.shape[-2 if not for_eval else 0] == len(ln)
    feats, labels = ev.extract_features(model, ev.TensorBatchLoader(imgs, batch, names, pids), print_freq=0, for_eval=False, group=g)
    rfeats, rlabels = ev.extract_features(model, ev.TensorBatchLoader(imgs, batch, names, pids), print_freq=0, for_eval=False)
    ok = ok and list(feats) == list(rfeats) == names and labels == rlabels
    ok = ok and all(len(feats[n]) == 3 and all(torch.equal(a, b) for a, b in zip(feats[n], rfeats[n])) for n in names)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_img,batch", [(2, 37, 4), (3, 10, 4), (3, 5, 4)])
def test_sharded_extraction_plumbing_gloo(world, n_img, batch):
    """extract_embeddings / extract_features(group=): contiguous batch blocks per rank + all-gather == the unsharded call
    (ragged last batch, fewer batches than ranks, loaders with and without shard())."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_extract_cpu_worker, args=(r, world, port, q, n_img, batch)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(60) for p in procs]
    assert res == [(r, True) for r in range(world)]


def test_loader_shards_are_balanced_by_image():
    """shard(rank, world) splits the IMAGES (sizes differ by at most one), not the batches: the bench's 12 936 source images in
    batches of 1000 over 8 ranks are 1617 per rank, not 2000 / 1000; listing() describes the whole loader without reading pixels."""
    from ssg_amd import evaluators as ev
    from ssg_amd.preprocessor import GpuBatchLoader
    imgs = torch.empty(12936, 0)
    full = ev.TensorBatchLoader(imgs, 1000)
    parts = [full.shard(r, 8) for r in range(8)]
    assert [p.num_items() for p in parts] == [1617] * 8 and [p.first for p in parts] == [1617 * r for r in range(8)]
    assert [len(p) for p in parts] == [2] * 8
    names, pids = full.listing()
    assert len(names) == 12936 and names[:2] == ["00000000", "00000001"] and pids[:3] == [0, 0, 0]
    assert sum((p.listing()[0] for p in parts), []) == names
    # round 5: a rank that holds only ITS block of the set (bench.py generates one block per rank): same description, resident slice
    whole = torch.arange(37 * 2, dtype=torch.float32).view(37, 2)
    lo, hi = ev.TensorBatchLoader(whole, 5).shard(1, 3).first, None
    for r in range(3):
        sh = ev.TensorBatchLoader(whole, 5).shard(r, 3)
        lo, hi = sh.first, sh.first + sh.count
        part = ev.TensorBatchLoader(whole[lo:hi].clone(), 5, count=37, base=lo)
        assert part.num_items() == 37 and part.listing() == ev.TensorBatchLoader(whole, 5).listing()
        mine = part.shard(r, 3)
        got = torch.cat([b[0] for b in mine]); names = sum((list(b[1]) for b in mine), [])
        assert torch.equal(got, whole[lo:hi]) and names == ["%08d" % i for i in range(lo, hi)]
        if r != 1:
            with pytest.raises(IndexError):
                list(part.shard(1, 3))
    ds = [("f%03d.jpg" % i, i % 5, i % 2) for i in range(37)]
    g = GpuBatchLoader(ds, root="/nonexistent", batch_size=8, decode="pillow")
    sub = [g.shard(r, 3) for r in range(3)]
    assert [x.num_items() for x in sub] == [13, 12, 12] and sum((x.listing()[0] for x in sub), []) == [d[0] for d in ds]
    assert g.listing()[1] == [d[1] for d in ds]


def _extract_gpu_worker(rank, world, port, q, n_img, batch, backend):
    import torch.distributed as dist
    import ssg_amd
    from ssg_amd import evaluators as ev
    torch.cuda.set_device(0)                       # before the process group (RCCL binds the communicator to the current device)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    imgs = torch.randn(n_img, 3, 256, 128, generator=torch.Generator().manual_seed(17))
    model = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=1, pretrained=False).cuda().eval()
    got, names, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, batch), for_eval=False, group=g)
    q.put((rank, got.cpu().numpy(), names))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])        # (8 processes each building their own ResNet-50 took 5 minutes of CPU; the 8-rank exchange pattern is covered by the pipeline test)
def test_sharded_extraction_matches_unsharded(world):
    """The embedding half of the multi-GPU path (selftraining.py:135,196-209 under nn.DataParallel in the reference): images
    sharded over `world` processes (gloo; they share the test box's single GPU), embeddings all-gathered -- bit-identical to the
    single-process extraction, in loader order, on every rank."""
    import ssg_amd
    from ssg_amd import evaluators as ev
    n_img, batch = 46, 4                               # 46 = 16 + 15 + 15 over three ranks: ragged shares, ragged last batches
    imgs = torch.randn(n_img, 3, 256, 128, generator=torch.Generator().manual_seed(17))
    model = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=1, pretrained=False).cuda().eval()
    ref, rnames, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, batch), for_eval=False)
    one, _, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, n_img), for_eval=False)
    assert torch.equal(ref, one), "an image's features must not depend on its batch"
    ref = ref.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_extract_gpu_worker, args=(r, world, port, q, n_img, batch, "gloo")) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=900) for _ in range(world)]
    [p.join(120) for p in procs]
    assert sorted(r[0] for r in res) == list(range(world))
    for r, got, names in res:
        assert names == rnames and got.shape == ref.shape == (3, n_img, 2048)
        assert np.array_equal(got, ref), "rank %d" % r


def _nccl_world1_worker(port, q):
    """the RCCL code path with one rank: every collective of the sharded pipeline executes on the nccl backend"""
    import torch.distributed as dist
    from types import SimpleNamespace
    import ssg_amd
    from ssg_amd import compute_dist, generate_selflabel, evaluators as ev, dist as sd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    g = dist.group.WORLD
    assert str(dist.get_backend(g)) == "nccl"
    out = {}
    N, Ns, d = 1531, 640, 96
    tgts = [torch.from_numpy(clustered(N, d, 5 + s)).to(dev) for s in range(2)]
    srcs = [torch.from_numpy(clustered(Ns, d, 60 + s, intra=0.7)).to(dev) for s in range(2)]
    for mode, no_rerank in (("rerank", False), ("norerank", True)):
        res = []
        for grp in (g, None):
            e_list, r_list = compute_dist(srcs, tgts, lambda_value=0.1, no_rerank=no_rerank, num_split=2, group=grp, grouping="shard")
            labels, clusters = generate_selflabel(e_list, r_list, 0, SimpleNamespace(no_rerank=no_rerank, rho=1.6e-3), [])
            hs = e_list if no_rerank else r_list
            res.append([(float(c.eps), l, h.M.cpu().numpy().view(np.uint16)) for c, l, h in zip(clusters, labels, hs)])
        out[mode] = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(*res))
    imgs = torch.randn(9, 3, 256, 128, generator=torch.Generator().manual_seed(17))
    model = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=1, pretrained=False).cuda().eval()
    a, na, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, 4), group=g)
    b, nb, _ = ev.extract_embeddings(model, ev.TensorBatchLoader(imgs, 4))
    out["extract"] = bool(torch.equal(a, b)) and na == nb
    t = torch.arange(12, dtype=torch.int32, device=dev).view(6, 2)
    out["gathers"] = bool(torch.equal(sd.gather_rows(t, g, 6), t)) and bool(torch.equal(sd.gather_varlen(t, g), t)) and any(k[0] == "nccl" and ok is True for k, ok in sd._FLAT_OK.items())
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_backend_world1_smoke():
    """torch.distributed's `nccl` backend IS RCCL on ROCm; the test box has one GPU, so this is a one-rank communicator --
    what it proves is that every collective of the sharded path (flat all-gathers of row tables / embeddings / ragged blocks,
    the eps all-reduces, all_gather_object, barrier) executes on RCCL with device tensors, before an 8-GPU node ever sees it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(120)
    assert out == {"rerank": True, "norerank": True, "extract": True, "gathers": True}, out


def _footprint_worker(rank, world, port, q, N, Ns, d):
    import hashlib
    import torch.distributed as dist
    from ssg_amd import rerank, cluster
    from ssg_amd import dist as sd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = dist.group.WORLD
    tgt = torch.from_numpy(clustered(N, d, 31)).to(dev); src = torch.from_numpy(clustered(Ns, d, 32, intra=0.7)).to(dev)
    lo, hi = sd.shard_bounds(N, rank, world)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    h = rerank.re_ranking_device(src, tgt, lambda_value=0.1, row0=lo, nrows=hi - lo, group=g, keep_euclid=False)
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    held = h.M.numel() * h.M.element_size()
    q.put((rank, float(eps), int(cnt), hashlib.sha256(lab.tobytes()).hexdigest(), int(peak), int(held), tuple(h.M.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_config4_sharded_footprint_and_result():
    """BASELINE configs[4]: N = 128 000 row-block-sharded over 8 ranks (here: 8 processes on the test box's one GPU, gloo).  Asserted,
    not assumed: a rank holds a 16 000 x 128 000 block of J' (4.1 GB) and, while the re-rank runs, the same block of D -- the peak
    of a rank's allocations stays below 8.2 GB + the sparse tables and the ranking hand-over records (the dense
    matrices are never gathered); eps, the element count and the labels equal the unsharded run's."""
    import hashlib
    from ssg_amd import rerank, cluster
    from ssg_amd.dist import shard_bounds
    N, Ns, d, world = 128000, 2000, 256, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_footprint_worker, args=(r, world, port, q, N, Ns, d)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=1500) for _ in range(world))
    [p.join(120) for p in procs]
    dev = torch.device("cuda", 0)
    tgt = torch.from_numpy(clustered(N, d, 31)).to(dev); src = torch.from_numpy(clustered(Ns, d, 32, intra=0.7)).to(dev)
    h = rerank.re_ranking_device(src, tgt, lambda_value=0.1, keep_euclid=False)
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
    want = hashlib.sha256(lab.tobytes()).hexdigest()
    block = 2.0 * (N // world) * N                     # bytes of one half row block
    for r, e, c, hsh, peak, held, shape in res:
        lo, hi = shard_bounds(N, r, world)
        assert shape == (hi - lo, N) and held == 2 * (hi - lo) * N
        assert (e, c, hsh) == (float(eps), int(cnt), want), "rank %d result differs from the unsharded run" % r
        assert peak < 2 * block + 2.5e9, "rank %d peak allocation %.2f GB (two half row blocks are %.2f GB)" % (r, peak / 1e9, 2 * block / 1e9)
    print("configs[4] sharded over 8 ranks: peak allocation per rank %.2f .. %.2f GB (D + J' row blocks: %.2f GB)" % (
        min(x[4] for x in res) / 1e9, max(x[4] for x in res) / 1e9, 2 * block / 1e9))


def _run_bench(world, extra_env=None, extra_args=()):
    """bench.py as the driver launches it (torchrun for world > 1), on a reduced problem; returns the parsed JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--gpus", str(world), "--steps", "1", "--warmup", "0", "--N", "4000", "--Ns", "2000", "--batch", "250", "--no-cpu-baseline", "--no-extras"]
    args += list(extra_args)
    env = dict(os.environ, OMP_NUM_THREADS="8", MKL_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    if world == 1:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_step_at_world_2_and_8_equals_world_1():
    """the bench's own step (sharded extraction -> all-gather of the embeddings -> row-block sharded re-rank -> eps ->
    DBSCAN) under torchrun at world 2 and 8 -- every rank on the one GPU of the test box (SSG_BENCH_SHARE_GPU=1: gloo collectives,
    the same product code path as RCCL) -- must produce the world-1 labels; the line reports the host syncs and the collectives of
    the sharded leg."""
    one = _run_bench(1)
    assert one["n_gpus"] == 1 and one["collectives"]["n"] == 0 and one["host_syncs"]["per_split"] <= 4
    for world in (2, 8):
        # round 6: the grouping leg has two forms (dist.choose_grouping).  'shard' = row blocks + all-gathers; at this reduced N the
        # default ('auto') picks 'replicate' -- every rank runs the whole leg, zero collectives in the grouping leg -- so both are run
        out = _run_bench(world, {"SSG_BENCH_SHARE_GPU": "1"}, ("--grouping", "shard"))
        assert out["n_gpus"] == world and out["config"]["N"] == 4000 and out["config"]["grouping_form"] == "shard"
        assert out["labels"] == one["labels"], (world, out["labels"], one["labels"])
        gathers = out["collectives"]["calls"].get("all_gather_into_tensor", 0) + out["collectives"]["calls"].get("all_gather", 0)
        assert out["collectives"]["n"] > 0 and 3 <= gathers <= 6, out["collectives"]
        assert out["host_syncs"]["world"] == world and out["host_syncs"]["per_split"] <= 3
        print("world %d sharded: %d collectives, %.1f MB received, %d host syncs per split, %.1f ms grouping" % (
            world, out["collectives"]["n"], out["collectives"]["bytes_received"] / 1e6, out["host_syncs"]["per_split"],
            out["rerank_dbscan_s_per_iter"] * 1e3))
    auto = _run_bench(2, {"SSG_BENCH_SHARE_GPU": "1"})
    assert auto["config"]["grouping_form"] == "replicate" and auto["collectives"]["n"] == 0, (auto["config"]["grouping_form"], auto["collectives"])
    assert auto["labels"] == one["labels"]
