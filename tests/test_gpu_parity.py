"""GPU suite (-m gpu): the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors of the reference.  Integer / half / label work is compared BIT-EXACT; there is
no floating-point tolerance anywhere in this file except where stated."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits, clustered, hard_clustered

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def L():
    from ssg_amd import _lib
    return _lib.lib()


def _sparse_to_dense(idx, val, nnz, N):
    idx = idx.cpu().numpy(); val = val.cpu().numpy().view(np.uint16); nnz = nnz.cpu().numpy()
    out = np.zeros((idx.shape[0], N), np.uint16)
    for i in range(idx.shape[0]):
        n = int(nnz[i])
        assert np.all(np.diff(idx[i, :n]) > 0), "sparse row %d not sorted by column" % i
        out[i, idx[i, :n]] = val[i, :n]
    return out


# ------------------------------------------------------------------ exhaustive half functions
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4])
def test_half_function_tables(which, L, dev, ora):
    from ssg_amd._lib import check, ptr, stream
    out = torch.empty(65536, dtype=torch.int16, device=dev)
    check(L.ssg_selftest_half_table(which, ptr(out), stream()), "selftest")
    got = out.cpu().numpy().view(np.uint16)
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    with np.errstate(all="ignore"):
        if which == 0:
            ref = ora.half_exp_table().view(np.uint16)
        elif which == 1:   # 1 - exp(-x)
            e = ora.half_exp_table().view(np.uint16)[(np.arange(65536) ^ 0x8000)].view(np.float16)
            ref = (np.float16(1) - e).view(np.uint16)
        elif which == 2:
            ref = np.power(allh, 2).view(np.uint16)
        elif which == 3:
            ref = (1 - allh / (2 - allh)).view(np.uint16)
        else:
            ref = np.sqrt(allh.astype(np.float64)).astype(np.float16).view(np.uint16)
    nan = np.isnan(got.view(np.float16)) & np.isnan(ref.view(np.float16))
    bad = np.nonzero((got != ref) & ~nan)[0]
    assert len(bad) == 0, [(hex(i), hex(got[i]), hex(ref[i])) for i in bad[:8]]


def test_half_binops_and_d2h(L, dev):
    from ssg_amd._lib import check, ptr, stream
    rng = np.random.default_rng(0)
    n = 1 << 20
    a = rng.integers(0, 0x7C00, n).astype(np.uint16); b = rng.integers(1, 0x7C00, n).astype(np.uint16)
    ta = torch.from_numpy(a.view(np.int16)).to(dev); tb = torch.from_numpy(b.view(np.int16)).to(dev)
    out = torch.empty(n, dtype=torch.int16, device=dev)
    fa, fb = a.view(np.float16), b.view(np.float16)
    with np.errstate(all="ignore"):
        refs = [fa + fb, fa / fb, fa * fb, fa - fb]
    for which, ref in enumerate(refs):
        check(L.ssg_selftest_half_binop(which, ptr(ta), ptr(tb), n, ptr(out), stream()), "binop")
        assert np.array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16)), which
    x = np.concatenate([rng.standard_normal(n) * 10.0 ** rng.integers(-9, 6, n), rng.random(n) * 4,
                        (rng.integers(0, 0x7C00, n).astype(np.uint16).view(np.float16).astype(np.float64)) * (1 + 2.0 ** -12)])
    tx = torch.from_numpy(x).to(dev)
    out = torch.empty(x.size, dtype=torch.int16, device=dev)
    check(L.ssg_selftest_d2h(ptr(tx), x.size, ptr(out), stream()), "d2h")
    with np.errstate(all="ignore"):
        assert np.array_equal(out.cpu().numpy().view(np.uint16), x.astype(np.float16).view(np.uint16))


def test_sort_u64(L, dev):
    from ssg_amd._lib import check, ptr, stream
    rng = np.random.default_rng(1)
    for n in (2048, 4096, 1 << 15, 1 << 18):
        a = rng.integers(0, 2 ** 62, n, dtype=np.int64)
        a[: n // 4] = a[0]
        t = torch.from_numpy(a).to(dev)
        check(L.ssg_sort_u64(ptr(t), n, stream()), "sort")
        assert np.array_equal(t.cpu().numpy(), np.sort(a))


# ------------------------------------------------------------------ pairwise distance (K3/K4)
@pytest.mark.parametrize("N,Ns,d", [(300, 77, 96), (129, 260, 40), (1024, 512, 256), (700, 900, 2048), (515, 1300, 160)])
def test_pairwise_vs_oracle(N, Ns, d, dev, ora):
    from ssg_amd import rerank
    tgt = clustered(N, d, 3); src = clustered(Ns, d, 4, intra=0.7)
    tgt[7] = tgt[2]                                     # exact duplicate rows -> exact zero distance
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), no_rerank=True)
    D = h.euclid.cpu().numpy()
    ref = ora.euclid(tgt)
    assert np.array_equal(bits(D), bits(ref))
    assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0) and D[7, 2] == 0
    rowmin = rerank.source_vector(rerank._as_dev_f32(src, dev), rerank._as_dev_f32(tgt, dev))                    # filter-and-refine
    rowmin_full = rerank.source_vector(rerank._as_dev_f32(src, dev), rerank._as_dev_f32(tgt, dev), exact_gemm=True)   # full fp64 Gram
    assert torch.equal(rowmin, rowmin_full)
    v_raw, v, mx = ora.source_vec(tgt, src)
    # rowmin holds half(d^2) bits; v_raw = 1-exp(-min) (monotone) -> compare through the finish kernel
    from ssg_amd._lib import check, ptr, stream, lib
    vv = torch.empty(N, dtype=torch.float16, device=dev); vm = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().ssg_source_vec_finish(ptr(rowmin), N, ptr(vv), ptr(vm), stream()), "finish")
    assert np.array_equal(bits(vv.cpu().numpy()), bits(v))
    assert int(vm.item()) == int(np.float16(mx).view(np.uint16))


# ------------------------------------------------------------------ full re-rank, stage by stage
RERANK = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "rerank_n*.npz")) + glob.glob(os.path.join(GOLDEN, "rerank_tiefree_*.npz")) +
                glob.glob(os.path.join(GOLDEN, "rerank_var_*.npz")))


def _check_rerank_golden(g, mode, dev, ora):
    from ssg_amd import rerank, cluster
    src, tgt = g["src"], g["tgt"]
    k1, k2, lam = int(g["k1"]), int(g["k2"]), float(g["lambda_value"])
    N = tgt.shape[0]
    msave = bool(g["memory_save"]) if "memory_save" in g.files else False      # MemorySave=True branch, rerank.py:49-59
    st = {}
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k1=k1, k2=k2, lambda_value=lam, stages=st,
                                 rank_mode=mode, memory_save=msave)
    oe, of, ost = ora.re_ranking(src, tgt, k1=k1, k2=k2, lambda_value=lam, MemorySave=msave, rank_mode=mode or rerank.default_rank_mode(),
                                 stages=True)
    assert np.array_equal(bits(st["D"].cpu().numpy()), bits(oe)), "original distance"
    assert np.array_equal(bits(st["v"].cpu().numpy()), bits(ost["v"])), "source vector"
    assert np.array_equal(st["rank"].cpu().numpy()[:, :g["rank"].shape[1]], g["rank"]), "initial rank (golden)"
    assert np.array_equal(st["rank"].cpu().numpy(), ost["rank"]), "initial rank (oracle, all columns)"
    assert np.array_equal(_sparse_to_dense(st["v_idx"], st["v_val"], st["v_nnz"], N), bits(ost["V"])), "V"
    assert np.array_equal(_sparse_to_dense(st["q_idx"], st["q_val"], st["q_nnz"], N), bits(ost["V_qe"])), "V_qe"
    assert np.array_equal(bits(st["Jp"].cpu().numpy()), bits(ost["jaccard_scaled"])), "scaled jaccard"
    final = h.final_dist().cpu().numpy()
    assert np.array_equal(final, of), "final_dist vs oracle"
    if "final" in g.files:
        assert np.array_equal(final, g["final"]), "final_dist vs reference golden"
        assert np.array_equal(bits(st["D"].cpu().numpy()), bits(g["euclid"]))
        assert np.array_equal(_sparse_to_dense(st["v_idx"], st["v_val"], st["v_nnz"], N), bits(g["V"])), "V vs reference golden"
        assert np.array_equal(_sparse_to_dense(st["q_idx"], st["q_val"], st["q_nnz"], N), bits(g["V_qe"])), "V_qe vs reference golden"
    eps, cnt, top = cluster.eps_rule(h, float(g["rho"]))
    assert (eps, cnt, top) == (float(g["eps"]), int(g["count"]), int(g["top_num"])), "eps rule"
    labels = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(h)
    assert np.array_equal(labels, g["labels"]), "DBSCAN labels (incl. numbering)"
    # the same through the materialised float64 matrix (sklearn drop-in path, mode 2)
    eps2, cnt2, top2 = cluster.eps_rule(final, float(g["rho"]))
    assert (eps2, cnt2, top2) == (eps, cnt, top)
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(final), g["labels"])


@pytest.mark.parametrize("name", RERANK)
def test_rerank_stages_vs_reference_golden(name, golden, dev, ora):
    """HIP path vs the reference's own outputs and vs the oracle at every stage boundary, bit-exact.
    '*_ref' fixtures come from the UNTOUCHED reference (np.argsort's unstable introsort tie order,
    reid/rerank.py:70): the default rank_mode must reproduce them.  '*_stable' fixtures come from the
    reference with argsort pinned to kind='stable' (opt-in rank_mode='stable').  Tie-free fixtures
    (untouched reference) must come out of both modes."""
    g = golden(name)
    if bool(g["stable"]):
        _check_rerank_golden(g, "stable", dev, ora)
    else:
        from ssg_amd import rerank
        assert rerank.default_rank_mode() == "introsort"
        _check_rerank_golden(g, None, dev, ora)          # default mode
        if bool(g["tie_free"]):
            _check_rerank_golden(g, "stable", dev, ora)


def test_query_expansion_guess_miss_is_redone(golden, dev, ora):
    """round 4: on one GPU the query expansion runs on a GUESSED longest V row (no host round trip) and reports a longer row through
    the status words the eps rule reads anyway.  Force a miss (guess = 4 entries): the eps rule must notice it with its first
    read-back, the Jaccard matrix must be rebuilt with the exact bound, the passes rerun, and everything equal the reference's
    outputs; the next guess must have grown; the exact-bound path (SSG_QE_GUESS=0) gives the same bits."""
    from ssg_amd import rerank, cluster
    g = golden("rerank_n256_l03_ref.npz")
    src, tgt = torch.from_numpy(g["src"]).to(dev), torch.from_numpy(g["tgt"]).to(dev)
    old = dict(rerank._QE_GUESS)
    try:
        rerank._QE_GUESS[(20, None)] = 4
        h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, validate=False)
        assert h._pending is not None and h._redo is not None
        stale = h.M.clone()
        eps, cnt, top = cluster.eps_rule(h, float(g["rho"]))
        assert h._pending is None and not torch.equal(stale, h.M), "the truncated matrix was not rebuilt"
        assert (eps, cnt, top) == (float(g["eps"]), int(g["count"]), int(g["top_num"]))
        assert np.array_equal(h.final_dist().cpu().numpy(), g["final"])
        assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), g["labels"])
        assert rerank._QE_GUESS[(20, None)] >= 32
        # the same miss caught by validate() (DBSCAN / final_dist as the first consumer)
        rerank._QE_GUESS[(20, None)] = 4
        h2 = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, validate=False)
        assert np.array_equal(h2.final_dist().cpu().numpy(), g["final"])
        # a sufficient guess: nothing is redone
        h3 = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, validate=False)
        before = h3.M.clone()
        h3.validate()
        assert torch.equal(before, h3.M) and np.array_equal(h3.final_dist().cpu().numpy(), g["final"])
    finally:
        rerank._QE_GUESS.clear(); rerank._QE_GUESS.update(old)
    os.environ["SSG_QE_GUESS"] = "0"
    try:
        h4 = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3)
    finally:
        del os.environ["SSG_QE_GUESS"]
    assert np.array_equal(h4.final_dist().cpu().numpy(), g["final"])


def test_rerank_second_stream_gives_the_one_stream_result(dev, monkeypatch):
    """round 5: on one GPU the source term runs on a second HIP stream beside the k-reciprocal kernels and is joined after the Jaccard rows
    (SSG_RERANK_OVERLAP, default on): J', the source vector, the sparse copy's floor word and the status words equal the one-stream
    call's bit for bit, repeatedly (a missing join would show as a stale or racing source vector), also when the handle is consumed
    right away by the fused eps rule + DBSCAN chain"""
    from ssg_amd import rerank, cluster
    tgt = hard_clustered(2500, 128, 41); src = hard_clustered(1100, 128, 42, intra=0.7)
    s_d, t_d = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    monkeypatch.setenv("SSG_RERANK_OVERLAP", "0")
    h0 = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3)
    ref = (h0.M.cpu().numpy().copy(), h0.v.cpu().numpy().copy(), int(h0.sparse["vmin"].item()))
    lab0 = cluster.eps_rule_dbscan(h0, 1.6e-3, min_samples=4)
    monkeypatch.setenv("SSG_RERANK_OVERLAP", "1")
    for _ in range(4):
        junk = torch.randn(1 << 22, device=dev)           # (main-stream work in front: the second stream has to wait for it)
        h1 = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3, validate=False)      # (no read-back between the join and the chain)
        floor1 = h1.sparse["vmin"]
        lab1 = cluster.eps_rule_dbscan(h1, 1.6e-3, min_samples=4)
        assert np.array_equal(bits(h1.v.cpu().numpy()), bits(ref[1]))
        assert np.array_equal(bits(h1.M.cpu().numpy()), bits(ref[0]))
        assert int(floor1.item()) == ref[2]
        assert lab1[:3] == lab0[:3] and np.array_equal(lab1[3], lab0[3])
        del junk


def test_eps_rule_dbscan_chain_equals_the_two_calls(dev, monkeypatch):
    """round 5: cluster.eps_rule_dbscan (eps rule -> region query -> components on the device, ONE read-back) against eps_rule followed
    by DBSCAN.fit on the same handle: eps, count, top, labels, core samples -- re-rank handles with and without the sparse copy, the half
    matrix of the no-rerank path, an uploaded float64 matrix, duplicate rows (zeros in the triangle: the device check fails, the
    two-call path answers), an edge list that is too small (the region query is redone), a missed query-expansion guess (the matrix is
    rebuilt and the chain runs again), a rho whose top is not the guessed one."""
    from ssg_amd import rerank, cluster
    rho = 1.6e-3

    def both(h, r=rho, ms=4):
        e0 = cluster.eps_rule(h, r)
        est = cluster.DBSCAN(eps=e0[0], min_samples=ms, metric="precomputed").fit(h)
        got = cluster.eps_rule_dbscan(h, r, min_samples=ms)
        assert bits(np.asarray(got[0])) == bits(np.asarray(e0[0])) if isinstance(e0[0], np.float16) else got[0] == e0[0], (got[0], e0[0])
        assert got[1:3] == e0[1:3]
        assert np.array_equal(got[3], est.labels_) and np.array_equal(got[4], est.core_sample_indices_)
        return got

    tgt = hard_clustered(3000, 96, 31); src = hard_clustered(900, 96, 32, intra=0.7)
    s_d, t_d = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    h = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3)
    assert h.sparse is not None
    g = both(h)
    assert g[3].max() > 3 and (g[3] < 0).any()
    both(h, ms=2)
    monkeypatch.setenv("SSG_SPARSE", "0")
    both(rerank.re_ranking_device(s_d, t_d, lambda_value=0.3))
    monkeypatch.delenv("SSG_SPARSE")
    he = rerank.re_ranking_device(s_d, t_d, no_rerank=True)                  # half matrix (mode 1): eps is a numpy float16
    ge = both(he)
    assert isinstance(ge[0], np.float16)
    both(h.final_dist().cpu().numpy())                                       # an uploaded float64 matrix (mode 2)
    # duplicates: exact zeros in the strict upper triangle -> count < N(N-1)/2, the guessed top may differ: the fallback answers
    td = tgt.copy(); td[10:40] = td[100:130]
    hd = rerank.re_ranking_device(s_d, torch.from_numpy(td).to(dev), no_rerank=True)
    gd = both(hd)
    assert gd[1] < 3000 * 2999 // 2
    # an edge list that is too small: large eps through rho (many neighbours per row)
    both(h, r=0.05)
    # the chain as the FIRST consumer of a handle whose query expansion ran on too small a guess
    old = dict(rerank._QE_GUESS)
    try:
        rerank._QE_GUESS[(20, None)] = 4
        hm = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3, validate=False)
        assert hm._pending is not None
        gm = cluster.eps_rule_dbscan(hm, rho)
        assert hm._pending is None and gm[0] == g[0] and np.array_equal(gm[3], g[3])
    finally:
        rerank._QE_GUESS.clear(); rerank._QE_GUESS.update(old)
    # host round trips of the chain: ONE
    import bench
    hc = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3, validate=False)
    cluster.eps_rule_dbscan(hc, rho)                                         # (tables cached)
    hc = rerank.re_ranking_device(s_d, t_d, lambda_value=0.3, validate=False)
    with bench.SyncCounter() as sc:
        cluster.eps_rule_dbscan(hc, rho)
    assert sc.n == 1, sc.n


def test_fused_sampling_and_check_launches_equal_the_separate_ones(dev):
    """round 6: `ssg_eps_sample_threshold` (each sampling level's last workgroup selects the threshold: two launches instead of four) and
    `ssg_eps_mean_check` (the a-posteriori checks in the tree kernel) against the separate entry points they replace in the chain -- the
    five threshold words and the six status words bit for bit, 30 times over (a workgroup that read the histogram before another one's
    atomics had landed would show as a different threshold), at two sizes, with other work in flight."""
    from ssg_amd import rerank
    from ssg_amd._lib import check, lib, ptr, stream
    L = lib()
    for N, d, seed in ((3000, 96, 31), (6000, 128, 33)):
        tgt = hard_clustered(N, d, seed); src = hard_clustered(900, d, seed + 1, intra=0.7)
        h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=0.3)
        args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
        stride, rho = max(1, N // 192), 1.6e-3
        hist = torch.zeros(2 * 4097, dtype=torch.int64, device=dev); thr = torch.zeros(5, dtype=torch.int64, device=dev)
        check(L.ssg_eps_sample_hist(*args, stride, None, ptr(hist[:4097]), stream()), "hist")
        check(L.ssg_eps_select_threshold(ptr(hist[:4097]), 1.3 * rho, ptr(thr), stream()), "select")
        check(L.ssg_eps_sample_hist(*args, stride, ptr(thr), ptr(hist[4097:]), stream()), "hist2")
        check(L.ssg_eps_refine_threshold(ptr(hist[4097:]), ptr(thr), stream()), "refine")
        ref_thr, ref_hist = thr.cpu().numpy().copy(), hist.cpu().numpy().copy()
        assert np.isfinite(np.uint32(ref_thr[0]).view(np.float32)) and ref_thr[1] > 1000
        for rep in range(30):
            junk = torch.randn(1 << 20, device=dev).sin_()                              # (other work in flight on the stream)
            z = torch.zeros(2 * 4097 + 5 + 1 + 1024, dtype=torch.int64, device=dev)
            check(L.ssg_eps_sample_threshold(*args, stride, 1.3 * rho, ptr(z[:8194]), ptr(z[8194:8199]), ptr(z[8199:8200]), ptr(z[8200:]), stream()), "fused")
            got = z.cpu().numpy()
            # the splitters derived from the histogram: ascending float64 bit patterns below the (coarse) threshold
            sp_ = got[8200:8200 + 1023].view(np.uint64)
            fin = sp_[sp_ != np.uint64(0xFFFFFFFFFFFFFFFF)].view(np.float64)
            assert len(fin) > 900 and np.all(np.diff(fin) >= 0) and fin[0] > 0 and fin[-1] <= float(np.uint32(ref_thr[0]).view(np.float32)) * 1.01, rep
            assert np.array_equal(got[:8194], ref_hist), rep
            assert np.array_equal(got[8194:8199], ref_thr), (rep, got[8194:8199], ref_thr)
            assert got[8199] == (int(got[8199]) & 0xffffffff) + ((int(got[8199]) >> 32) << 32) and (int(got[8199]) & 0xffffffff) == (int(got[8199]) >> 32) > 0   # both tickets = the grid size
            del junk
        # the checks inside the tree kernel: same status words and the same eps as the two launches
        top = int(np.round(rho * (N * (N - 1) // 2)))
        n_cap = 1 << 20
        buf = torch.empty(n_cap, dtype=torch.int64, device=dev); cursor = torch.zeros(3, dtype=torch.int64, device=dev)
        check(L.ssg_eps_compact_below(*args, ptr(thr), ptr(buf), n_cap, ptr(cursor), stream()), "compact")
        check(L.ssg_sort_u64_dev(ptr(buf), n_cap, ptr(cursor), stream()), "sort")
        wsb = int(L.ssg_eps_mean_workspace_bytes(top)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        check(L.ssg_eps_mean_prepare(top, ptr(ws), wsb, stream()), "prepare")
        outs = []
        for fused in (False, True):
            eps2 = torch.zeros(2, dtype=torch.float64, device=dev); st6 = torch.zeros(6, dtype=torch.int64, device=dev)
            if fused:
                check(L.ssg_eps_mean_check(ptr(buf), top, 0, ptr(ws), wsb, ptr(eps2), ptr(cursor), ptr(thr), rho, N * (N - 1) // 2, n_cap, ptr(st6), None, stream()), "mean_check")
            else:
                check(L.ssg_eps_mean_run(ptr(buf), top, 0, ptr(ws), wsb, ptr(eps2), stream()), "mean")
                check(L.ssg_eps_check(ptr(buf), ptr(cursor), ptr(thr), rho, N * (N - 1) // 2, top, n_cap, ptr(eps2), ptr(st6), None, stream()), "check")
            outs.append((eps2.cpu().numpy().view(np.int64).copy(), st6.cpu().numpy().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][1][0] == 1
        # ... and a failing check (a top that is not the tree's) poisons eps in both forms
        eps2 = torch.zeros(2, dtype=torch.float64, device=dev); st6 = torch.zeros(6, dtype=torch.int64, device=dev)
        check(L.ssg_eps_mean_check(ptr(buf), top, 0, ptr(ws), wsb, ptr(eps2), ptr(cursor), ptr(thr), 2.0 * rho, N * (N - 1) // 2, n_cap, ptr(st6), None, stream()), "mean_check")
        assert int(st6[0].item()) == 0 and np.isnan(eps2[0].item())


def test_sort_u64_dev_sorts_the_device_count(dev):
    """ssg_sort_u64_dev: buf[0 .. *n_dev) sorted inside a buffer of n_cap entries for counts around the powers of two, 0 and n_cap"""
    from ssg_amd import _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib()
    n_cap = 1 << 15
    g = torch.Generator().manual_seed(3)
    for n in (0, 1, 5, 2047, 2048, 2049, 4096, 5000, 8193, 20000, n_cap - 1, n_cap):
        keys = torch.randint(0, 1 << 62, (n_cap,), generator=g, dtype=torch.int64)
        buf = keys.to(dev)
        nd = torch.tensor([n, 7, 9], dtype=torch.int64, device=dev)
        check(L.ssg_sort_u64_dev(ptr(buf), n_cap, ptr(nd), stream()), "ssg_sort_u64_dev")
        assert torch.equal(buf[:n].cpu(), torch.sort(keys[:n]).values), n


def test_samplesort_u64_dev_sorts_the_device_count(dev):
    """ssg_samplesort_u64_dev (5 launches, any capacity): buf[0 .. *n_dev) sorted for counts around the chunk / sample sizes, on uniform
    keys, on double bit patterns crowded against one end, on a few hundred distinct values (half-valued distances: every key a duplicate),
    on all-equal keys and on an already sorted list; the fail word stays 0; the tail past the count is untouched"""
    from ssg_amd import _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    for n_cap in (1, 777, 1 << 15, 300001):
        wsb = int(L.ssg_samplesort_u64_workspace_bytes(n_cap))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fail = torch.full((1,), 5, dtype=torch.int64, device=dev)
        makers = {
            "uniform": lambda: torch.randint(0, 1 << 62, (n_cap,), generator=g, dtype=torch.int64),
            "crowded doubles": lambda: (0.9 - 0.9 * torch.rand(n_cap, generator=g, dtype=torch.float64) ** 8).view(torch.int64),
            "few distinct": lambda: torch.randint(0, 300, (n_cap,), generator=g, dtype=torch.int64) * 1000003,
            "all equal": lambda: torch.full((n_cap,), 424242, dtype=torch.int64),
            "sorted": lambda: torch.arange(n_cap, dtype=torch.int64) * 3,
        }
        for name, mk in makers.items():
            for n in sorted({0, 1, 2, 5, 4095, 4096, 4097, 8191, 8192, 8193, 20000, n_cap - 1, n_cap}):
                if n < 0 or n > n_cap:
                    continue
                keys = mk()
                buf = keys.to(dev)
                nd = torch.tensor([n, 7, 9], dtype=torch.int64, device=dev)
                check(L.ssg_samplesort_u64_dev(ptr(buf), n_cap, ptr(nd), ptr(ws), wsb, ptr(fail), stream()), "ssg_samplesort_u64_dev")
                out = buf.cpu()
                assert int(fail.item()) == 0, (name, n_cap, n)
                assert torch.equal(out[:n], torch.sort(keys[:n]).values), (name, n_cap, n)
                assert torch.equal(out[n:], keys[n:]), (name, n_cap, n)
    # round 6: the second geometry (4095 splitters out of a sample of 16 384, buckets of up to 16 384 keys in LDS) on larger lists
    for n_cap in (1, 40000, 1 << 21):
        wsb = int(L.ssg_samplesort_u64_big_workspace_bytes(n_cap))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fail = torch.full((1,), 5, dtype=torch.int64, device=dev)
        makers = {
            "uniform": lambda: torch.randint(0, 1 << 62, (n_cap,), generator=g, dtype=torch.int64),
            "crowded doubles": lambda: (0.9 - 0.9 * torch.rand(n_cap, generator=g, dtype=torch.float64) ** 8).view(torch.int64),
            "few distinct": lambda: torch.randint(0, 300, (n_cap,), generator=g, dtype=torch.int64) * 1000003,
            "all equal": lambda: torch.full((n_cap,), 424242, dtype=torch.int64),
            "sorted": lambda: torch.arange(n_cap, dtype=torch.int64) * 3,
        }
        for name, mk in makers.items():
            for n in sorted({0, 1, 3, 16383, 16384, 16385, 65537, n_cap - 1, n_cap}):
                if n < 0 or n > n_cap:
                    continue
                keys = mk()
                buf = keys.to(dev)
                nd = torch.tensor([n, 7, 9], dtype=torch.int64, device=dev)
                check(L.ssg_samplesort_u64_big_dev(ptr(buf), n_cap, ptr(nd), ptr(ws), wsb, ptr(fail), stream()), "ssg_samplesort_u64_big_dev")
                out = buf.cpu()
                assert int(fail.item()) == 0, (name, n_cap, n)
                assert torch.equal(out[:n], torch.sort(keys[:n]).values), (name, n_cap, n)
                assert torch.equal(out[n:], keys[n:]), (name, n_cap, n)
    # a count above the capacity is clamped (the eps check reports it)
    n_cap = 5000
    wsb = int(L.ssg_samplesort_u64_workspace_bytes(n_cap)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    keys = torch.randint(0, 1 << 40, (n_cap,), generator=g, dtype=torch.int64); buf = keys.to(dev)
    nd = torch.tensor([n_cap + 99], dtype=torch.int64, device=dev); fail = torch.zeros(1, dtype=torch.int64, device=dev)
    check(L.ssg_samplesort_u64_dev(ptr(buf), n_cap, ptr(nd), ptr(ws), wsb, ptr(fail), stream()), "ssg_samplesort_u64_dev")
    assert torch.equal(buf.cpu(), torch.sort(keys).values)


def test_samplesort_oversized_bucket_paths(dev):
    """a distribution the strided sample cannot see: 30 000 distinct keys hidden in ONE stretch between two sampled positions' values
    (every sampled key is one of two values) -> buckets over the LDS size are ranked out of global memory, over 16384 keys raise the
    fail word (and `ssg_eps_check` turns that into ok = 0)"""
    from ssg_amd import _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib()
    n = 1 << 16                                       # sample stride 16: positions 0, 16, 32, ...
    for hidden, want_fail in ((5000, 0), (30000, 1)):
        keys = torch.full((n,), 10, dtype=torch.int64)
        keys[n // 2:] = 1 << 40
        pos = torch.arange(hidden) * 2 + 1            # odd positions: never sampled
        keys[pos] = 1000 + torch.randperm(hidden, generator=torch.Generator().manual_seed(1)).to(torch.int64)
        buf = keys.to(dev)
        wsb = int(L.ssg_samplesort_u64_workspace_bytes(n)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        nd = torch.tensor([n], dtype=torch.int64, device=dev); fail = torch.zeros(1, dtype=torch.int64, device=dev)
        check(L.ssg_samplesort_u64_dev(ptr(buf), n, ptr(nd), ptr(ws), wsb, ptr(fail), stream()), "ssg_samplesort_u64_dev")
        assert int(fail.item()) == want_fail
        if not want_fail:
            assert torch.equal(buf.cpu(), torch.sort(keys).values)
        else:
            assert torch.equal(torch.sort(buf.cpu()).values, torch.sort(keys).values)      # still a permutation of the keys


def test_range_stats_kernel(dev):
    """ssg_range_stats_f32 (one launch for the four value ranges the host decides on) against torch: maxima exact, norms upper bounds
    within 2e-5 relative; NaN propagates."""
    from ssg_amd import rerank
    g = torch.Generator().manual_seed(5)
    a = torch.randn(333, 160, generator=g).to(dev) * 0.3; b = torch.randn(77, 160, generator=g).to(dev)
    s = rerank.range_stats(a, b)
    assert s[0] == float(a.abs().max()) and s[1] == float(b.abs().max())
    for got, ref in ((s[2], float(a.double().norm(dim=1).max())), (s[3], float(b.double().norm(dim=1).max()))):
        assert ref <= got <= ref * (1 + 2e-5)
    s1 = rerank.range_stats(a)
    assert s1[0] == s[0] and s1[1] == 0.0 and s1[3] == 0.0
    a[7, 3] = float("nan")
    assert np.isnan(rerank.range_stats(a, b)[0])


def _rank_rows(keys, K, dev, mode="introsort", force_arena=False):
    """rows of order keys (uint16, < 0x3c00) -> device ranking of half(D / 1.0) with D = the keys as half bit patterns"""
    from ssg_amd import rerank
    D = torch.from_numpy(np.ascontiguousarray(keys).view(np.int16)).to(dev).view(torch.float16)
    rowmax = torch.full((keys.shape[0],), 0x3C00, dtype=torch.int32, device=dev)
    return rerank.initial_rank(D, rowmax, keys.shape[1], keys.shape[0], K, mode, force_arena=force_arena).cpu().numpy()


@pytest.mark.parametrize("N", [2, 16, 17, 18, 63, 64, 65, 100, 1000, 1024, 1025, 1027, 1028, 1091, 2500, 5000, 16000, 16522, 30011, 36000, 40000, 70001])
def test_introsort_rank_vs_numpy_argsort(N, dev, ora):
    """csrc/topk_intro.hip against np.argsort's default kind on the same half rows (the oracle's sequential restatement of
    numpy's aquicksort is checked against np.argsort right here as well): tie-heavy, constant, sorted, reversed and
    tie-free rows; row blocks that start at unaligned addresses; N > 36 k runs from the global arena."""
    rng = np.random.default_rng(N)
    rows = []
    for nv in (1, 2, 3, 7, 40, 300, 5000):
        rows.append(rng.integers(0, nv, N))
    rows.append(np.sort(rng.integers(0, 50, N)))
    rows.append(np.sort(rng.integers(0, 50, N))[::-1])
    rows.append(rng.permutation(N) % 15000)
    rows.append(np.arange(N) % 15000)
    rows.append((N - 1 - np.arange(N)) % 15000)
    keys = np.stack(rows).astype(np.uint16)
    if N > 20000:
        keys = keys[[1, 4, 6, 8, 9]]
    for K in sorted({1, min(21, N), min(64, N)}):
        got = _rank_rows(keys, K, dev)
        for r in range(keys.shape[0]):
            ref = np.argsort(keys[r].view(np.float16))[:K]
            if N <= 5000:
                assert np.array_equal(ora.argsort_half(keys[r].view(np.float16))[:K], ref)
            assert np.array_equal(got[r], ref), (N, K, r)
    if N <= 5000:      # the global-arena variant of the kernel on rows that would fit in LDS
        got = _rank_rows(keys, min(21, N), dev, force_arena=True)
        for r in range(keys.shape[0]):
            assert np.array_equal(got[r], np.argsort(keys[r].view(np.float16))[:min(21, N)]), (N, r, "arena")
    # stable mode on the same rows == np.argsort(kind='stable')
    K = min(21, N)
    got = _rank_rows(keys, K, dev, "stable")
    for r in range(keys.shape[0]):
        assert np.array_equal(got[r], np.argsort(keys[r].view(np.float16), kind="stable")[:K]), (N, r)


@pytest.mark.parametrize("N,cap", [(5000, 2048), (16523, 2400), (40000, 3000), (40000, 0), (70001, 9000)])
def test_introsort_streamed_levels_vs_numpy_argsort(N, cap, dev, monkeypatch):
    """round 5: rows that do not fit in LDS take their first levels as OUT-OF-PLACE streamed partitions (csrc/topk_intro.hip
    stream_partition: the row of D is only read, the left child is written once).  A small SSG_INTRO_STREAM_CAP forces several
    streamed levels (through the two global buffers, then into LDS) on rows of any length: np.argsort's order must come out for
    tie-heavy, constant, sorted, reversed, tie-free and adversarial rows, unaligned row starts (odd N), K = 1 / 21 / 64 -- including
    rows whose first pivot lands inside [0, K) (flagged and redone by the in-place kernel) -- and equal the in-place arena path."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from antiqsort import killer_keys
    rng = np.random.default_rng(N + cap)
    rows = [rng.integers(0, nv, N) for nv in (1, 2, 7, 300, 5000)]
    rows.append(np.sort(rng.integers(0, 50, N)))
    rows.append(np.sort(rng.integers(0, 50, N))[::-1])
    rows.append(rng.permutation(N) % 15000)
    bad = np.full(N, 1000); bad[0] = 0; bad[(N - 1) >> 1] = 0; bad[N - 1] = 1       # the first pivot (key 0) lands at the front: both children matter
    rows.append(bad)
    bad2 = rng.integers(5, 9000, N); bad2[[0, (N - 1) >> 1, N - 1]] = (3, 2, 4); bad2[rng.integers(1, N - 2, 9)] = 1   # pivot key 3 with ~10 smaller keys: pi ~ 10 < 21
    rows.append(bad2)
    for kq in (21, 64):        # the first pivot lands EXACTLY at column K - 1: it is the last output itself (no child holds it)
        row = rng.integers(100, 9000, N); row[[0, (N - 1) >> 1, N - 1]] = (50, 70, 90)
        row[rng.choice(np.arange(1, N - 1)[np.arange(1, N - 1) != ((N - 1) >> 1)], kq - 2, replace=False)] = rng.integers(1, 69, kq - 2)   # kq - 1 keys below the pivot (70)
        rows.append(row)
    kk = killer_keys(min(N, 6000)); rows.append(np.concatenate([kk, np.full(N - len(kk), 20000)]))
    keys = np.stack(rows).astype(np.uint16)
    if cap:
        monkeypatch.setenv("SSG_INTRO_STREAM_CAP", str(cap))
    force = N <= 36000
    for K in (1, 21, 64):
        got = _rank_rows(keys, K, dev, force_arena=force)
        for r in range(keys.shape[0]):
            assert np.array_equal(got[r], np.argsort(keys[r].view(np.float16))[:K]), (N, cap, K, r)
    monkeypatch.setenv("SSG_INTRO_STREAM", "0")
    old = _rank_rows(keys, 21, dev, force_arena=force)
    monkeypatch.delenv("SSG_INTRO_STREAM")
    assert np.array_equal(old, _rank_rows(keys, 21, dev, force_arena=force))
    # the LAST row of a matrix with an odd number of halves: its last element shares a dword with the 2 bytes behind the buffer (the row is
    # read through a bounds-checked buffer resource whose range check is per dword) -- a small key there must still rank first
    last = keys[:3].copy(); last[2, :] = 9000; last[2, N - 1] = 1; last[2, N - 2] = 2
    for K in (1, 21):
        got = _rank_rows(last, K, dev, force_arena=force)
        assert np.array_equal(got[2], np.argsort(last[2].view(np.float16))[:K]), (N, cap, K, "last row")
        if N <= 36000:
            assert np.array_equal(_rank_rows(last, K, dev)[2], got[2])


@pytest.mark.parametrize("N,d", [(6000, 64), (20000, 128)])
def test_introsort_streamed_levels_on_real_distance_rows(N, d, dev, monkeypatch):
    """the streamed levels on rows of a REAL original-distance matrix (row maxima != 1: keys are half(raw / rowmax), ties as the
    pipeline produces them) for several LDS capacities == the rows-in-LDS kernel (which the reference goldens pin) == np.argsort on a
    sample of rows.  (N = 20 000 at a capacity of 12 000 entries has a row whose first pivot lands exactly at column K - 1: the case
    that exposed a missing output column during development.)"""
    from ssg_amd import rerank, _lib
    tgt = rerank._as_dev_f32(clustered(N, d, 1), dev)
    D, rowmax, _ = rerank._original_distance(_lib.lib(), tgt, 0, N, float(tgt.abs().max()), _lib.stream())
    want = rerank.initial_rank(D, rowmax, N, N, 21, "introsort")
    for cap in (2048, 5000, 12000):
        monkeypatch.setenv("SSG_INTRO_STREAM_CAP", str(cap))
        got = rerank.initial_rank(D, rowmax, N, N, 21, "introsort", force_arena=True)
        assert torch.equal(got, want), (N, cap, int((got != want).any(dim=1).sum()))
    monkeypatch.delenv("SSG_INTRO_STREAM_CAP")
    rows = np.random.default_rng(0).choice(N, 40, replace=False)
    Dh = D[torch.from_numpy(rows).to(dev)].cpu().numpy(); rm = rowmax.cpu().numpy().astype(np.uint16).view(np.float16)
    w = want.cpu().numpy()
    for q, r in enumerate(rows):
        key = (Dh[q].astype(np.float32) / np.float32(rm[r])).astype(np.float16)
        assert np.array_equal(np.argsort(key)[:21], w[r]), r


def test_introsort_rank_without_workspace(dev):
    """the pre-round-3 calling convention (ws = NULL for rows that fit in LDS) still works -- the replay then runs unsplit
    in the single-launch kernel -- and gives the same ranking as the split (workspace) path."""
    from ssg_amd import _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib()
    rng = np.random.default_rng(4)
    N, K = 6000, 21
    keys = np.stack([rng.integers(0, nv, N) for nv in (3, 40, 700, 12000)]).astype(np.uint16)
    D = torch.from_numpy(keys.view(np.int16)).to(dev).view(torch.float16)
    rowmax = torch.full((4,), 0x3C00, dtype=torch.int32, device=dev)
    rank = torch.empty((4, K), dtype=torch.int32, device=dev)
    check(L.ssg_topk_rank_introsort(ptr(D), ptr(rowmax), N, 4, K, ptr(rank), None, 0, stream()), "ssg_topk_rank_introsort(ws=NULL)")
    got = rank.cpu().numpy()
    for r in range(4):
        assert np.array_equal(got[r], np.argsort(keys[r].view(np.float16))[:K]), r
    assert np.array_equal(got, _rank_rows(keys, K, dev))


@pytest.mark.parametrize("n", [100, 300, 1000, 3000])
def test_introsort_rank_heapsort_fallback(n, dev):
    """keys built by an adversary against median-of-3 quicksort (tools/antiqsort.py): numpy's argsort runs out of its
    depth budget and heapsorts a range that reaches into the first 64 columns -- so must the device kernel."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from antiqsort import killer_keys
    key = killer_keys(n)
    keys = np.stack([key, key[::-1].copy()])
    for K in (21, 64):
        for arena in (False, True):     # rows in LDS / rows in the global arena (what N > 36 k uses)
            got = _rank_rows(keys, K, dev, force_arena=arena)
            for r in range(2):
                assert np.array_equal(got[r], np.argsort(keys[r].view(np.float16))[:K]), (n, K, r, arena)


def test_reranking_dropin_signature(golden, dev):
    """reid/rerank.py:27 call surface: numpy in, (float16 [N,N], float64 [N,N]) out."""
    from ssg_amd import re_ranking, DBSCAN
    g = golden("rerank_n256_l03_ref.npz")          # the untouched reference's output (default tie order)
    e, f = re_ranking(g["src"], g["tgt"], k1=20, k2=6, lambda_value=0.3)
    assert e.dtype == np.float16 and f.dtype == np.float64 and f.shape == (256, 256)
    assert np.array_equal(np.asarray(f), g["final"]) and np.array_equal(bits(e), bits(g["euclid"]))
    lab = DBSCAN(eps=float(g["eps"]), min_samples=4, metric="precomputed", n_jobs=8).fit_predict(f)
    assert np.array_equal(lab, g["labels"])
    gm = golden("rerank_var_msave_ref.npz")        # MemorySave=True numerics (rerank.py:49-59), Minibatch accepted
    em, fm = re_ranking(gm["src"], gm["tgt"], k1=20, k2=6, lambda_value=0.1, MemorySave=True, Minibatch=40)
    assert np.array_equal(np.asarray(fm), gm["final"]) and np.array_equal(bits(em), bits(gm["euclid"]))
    with pytest.raises(ValueError):
        re_ranking(g["src"], g["tgt"], k1=64, k2=6)
    gs = golden("rerank_n256_l03_stable.npz")      # opt-in canonical (value, index) order
    _, fs = re_ranking(gs["src"], gs["tgt"], k1=20, k2=6, lambda_value=0.3, rank_mode="stable")
    assert np.array_equal(np.asarray(fs), gs["final"])
    e2, none = re_ranking(g["src"], g["tgt"], no_rerank=True)
    assert none is None and np.array_equal(bits(e2), bits(e))


@pytest.mark.parametrize("name", ["norerank_n256.npz", "norerank_n1024.npz"])
def test_norerank_path(name, golden, dev):
    from ssg_amd import rerank, cluster
    g = golden(name)
    tgt = torch.from_numpy(g["tgt"]).to(dev)
    h = rerank.re_ranking_device(tgt[:8], tgt, no_rerank=True)
    eps, cnt, top = cluster.eps_rule(h, float(g["rho"]))
    assert np.float16(eps).view(np.uint16) == int(g["eps_bits"]) and cnt == int(g["count"]) and top == int(g["top_num"])
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
    assert np.array_equal(lab, g["labels"])


def test_dbscan_cases(golden, dev):
    from ssg_amd import DBSCAN
    g = golden("dbscan_cases.npz")
    for D, eps, lab in zip(g["D"], g["eps"], g["labels"]):
        est = DBSCAN(eps=float(eps), min_samples=4, metric="precomputed").fit(D)
        assert np.array_equal(est.labels_, lab)
        counts = (D <= eps).sum(axis=1)
        assert np.array_equal(est.core_sample_indices_, np.nonzero(counts >= 4)[0])


def test_selftraining_surface_and_edge_cases(dev, ora):
    """compute_dist -> generate_selflabel (selftraining.py:255-313) on the fused device path,
    3 splits, ragged N (not a multiple of any tile), duplicates; vs the oracle."""
    from types import SimpleNamespace
    from ssg_amd import compute_dist, generate_selflabel
    N, Ns, d = 777, 345, 72
    tgts = [clustered(N, d, 40 + s) for s in range(3)]
    srcs = [clustered(Ns, d, 50 + s, intra=0.7) for s in range(3)]
    tgts[0][11] = tgts[0][10]
    e_list, r_list = compute_dist([torch.from_numpy(s) for s in srcs], [torch.from_numpy(t) for t in tgts], lambda_value=0.1,
                                  no_rerank=False, num_split=2)
    args = SimpleNamespace(no_rerank=False, rho=1.6e-3)
    labels, clusters = generate_selflabel(e_list, r_list, 0, args, [])
    oe, orr = ora.compute_dist(srcs, tgts, 0.1, False)
    olabels, oeps = ora.generate_selflabel(oe, orr, 0, 1.6e-3, False)
    for s in range(3):
        assert clusters[s].eps == oeps[s]
        assert np.array_equal(labels[s], olabels[s])
    # iteration 1 reuses the cached estimators (eps frozen, selftraining.py:297-298)
    labels2, clusters2 = generate_selflabel(e_list, r_list, 1, args, clusters)
    assert clusters2 is clusters and all(np.array_equal(a, b) for a, b in zip(labels, labels2))


def test_selftraining_functions_vs_the_references_own(dev, golden):
    """tests/golden/selftraining_ref.npz = outputs of the reference's OWN compute_dist / generate_selflabel / generate_dataloader
    (selftraining.py:255-331 imported by tools/make_golden.py, two iterations, 3 splits, N = 200): the product's functions of the
    same names give the same eps per split (iteration 0), the same labels incl. numbering (iteration 1 on the CACHED estimators:
    eps frozen, selftraining.py:297-298), the same final_dist bytes and the same joined dataset (a6, a8, a9, a10)."""
    import hashlib
    from types import SimpleNamespace
    from ssg_amd import compute_dist, generate_selflabel
    from ssg_amd.selftraining import generate_dataset
    g = golden("selftraining_ref.npz")
    N, S1 = int(g["N"]), int(g["splits"])
    args = SimpleNamespace(no_rerank=False, rho=float(g["rho"]))
    trainval = [("img_%05d_c%d.jpg" % (i, i % 6), i // 16, i % 6) for i in range(N)]
    clusters = []
    for it in range(2):
        src = [torch.from_numpy(g["src_%d_%d" % (it, s)]) for s in range(S1)]
        tgt = [torch.from_numpy(g["tgt_%d_%d" % (it, s)]) for s in range(S1)]
        e_list, r_list = compute_dist(src, tgt, float(g["lambda_value"]), False)
        assert e_list == [[]] * S1
        labels, clusters = generate_selflabel(e_list, r_list, it, args, clusters)
        assert len(clusters) == S1
        for s in range(S1):
            assert clusters[s].eps == float(g["eps_%d" % s]), (it, s)
            assert np.array_equal(np.asarray(labels[s]).astype(np.int64), g["labels_%d_%d" % (it, s)]), (it, s)
            f = r_list[s].final_dist().cpu().numpy()
            assert f.dtype == np.float64 and hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(g["sha_final_%d_%d" % (it, s)]), (it, s)
        ds = generate_dataset(trainval, labels, iter_n=it)
        assert [int(f[4:9]) for f, _, _ in ds] == g["kept_%d" % it].tolist() and all(c == 0 for _, _, c in ds)
        assert np.array_equal(np.array([[int(x) for x in lab] for _, lab, _ in ds], np.int64).reshape(len(ds), S1), g["kept_labels_%d" % it])


def test_nan_path_is_raised(dev):
    """reid/rerank.py:40 divides by max(source_dist_vec)==0 -> NaN; the build raises instead."""
    from ssg_amd import rerank
    x = clustered(64, 32, 1)
    with pytest.raises(rerank.ReRankNaNError):
        rerank.re_ranking_device(torch.from_numpy(x).to(dev), torch.from_numpy(x).to(dev))   # src == tgt -> every d2 == 0


@pytest.mark.parametrize("N", [4096])
def test_medium_size_vs_oracle(N, dev, ora):
    """Largest size the oracle finishes in seconds; full pipeline labels + distances bit-exact."""
    from ssg_amd import rerank, cluster
    d = 128
    tgt = clustered(N, d, 77); src = clustered(N // 2, d, 78, intra=0.7)
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=0.1)
    oe, of = ora.re_ranking(src, tgt, lambda_value=0.1)
    assert np.array_equal(bits(h.euclid.cpu().numpy()), bits(oe))
    assert np.array_equal(h.final_dist().cpu().numpy(), of)
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    oeps, ocnt, otop = ora.eps_rule(of, 1.6e-3)
    assert (eps, cnt, top) == (oeps, ocnt, otop)
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), ora.dbscan(of, oeps, 4))


def test_full_size_properties(dev):
    """BASELINE size N=16000, d=2048: size-independent properties (the oracle cannot run here)."""
    from ssg_amd import rerank, cluster
    N, d = 16000, 2048
    tgt = torch.from_numpy(clustered(N, d, 1)).to(dev); src = torch.from_numpy(clustered(8192, d, 2, intra=0.7)).to(dev)
    h = rerank.re_ranking_device(src, tgt, lambda_value=0.1)
    D, Jp = h.euclid, h.M
    assert torch.equal(D, D.T) and bool((torch.diagonal(D) == 0).all())
    assert torch.equal(Jp, Jp.T), "J' must be exactly symmetric"
    assert bool((Jp >= 0).all()) and bool((Jp.float() <= 1.0).all())
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    assert top == int(np.round(1.6e-3 * cnt)) and 0 < eps < 1.5
    est = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit(h)
    lab = est.labels_
    # cluster ids are numbered by their smallest core index (dbscan_inner visiting order)
    ids = lab[lab >= 0]
    core = est.core_sample_indices_
    first_core = {}
    for i in core:
        first_core.setdefault(int(lab[i]), int(i))
    order = [first_core[l] for l in sorted(first_core)]
    assert order == sorted(order) and sorted(first_core) == list(range(len(first_core)))
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), lab)
    # synthetic identities (16 per id) are recovered: every cluster is pure
    truth = np.arange(N) % (N // 16)
    for l in np.unique(ids)[:200]:
        assert len(np.unique(truth[lab == l])) == 1


# ------------------------------------------------------------------ embedding (K1/K2), fp32
# Floating-point kernels: tolerance stated per test (the torch fp32 CPU reference itself is
# only reproducible to ~1e-6 across BLAS/oneDNN kernel choices).
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, stride, pad, residual, relu
    (3, 16, 8, 64, 64, 1, 1, 0, False, True),
    (2, 16, 8, 64, 256, 1, 1, 0, True, True),
    (2, 16, 8, 128, 128, 3, 1, 1, False, True),
    (2, 16, 8, 128, 128, 3, 2, 1, False, True),
    (2, 16, 8, 256, 512, 1, 2, 0, False, False),
    (5, 9, 7, 64, 64, 3, 1, 1, False, True),        # ragged M (not a tile multiple)
    (2, 32, 16, 3, 64, 7, 2, 3, False, True),       # stem
])
def test_conv_vs_torch(cfg, L, dev):
    from ssg_amd._lib import check, ptr, stream
    B, H, W, Cin, Cout, k, stride, pad, use_res, relu = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride, pad)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    if Cin == 3:
        xin = torch.cat([x, torch.zeros(B, 1, H, W)], 1).permute(0, 2, 3, 1).contiguous()
        wk = torch.nn.functional.pad(w.permute(0, 2, 3, 1), (0, 1)).reshape(Cout, k * k * 4)
        wk = torch.nn.functional.pad(wk, (0, 32 * ((k * k + 7) // 8) - k * k * 4)).contiguous()
        cin = 4
    else:
        from ssg_amd.resnet import pack_weight_khwc
        xin = x.permute(0, 2, 3, 1).contiguous(); wk = pack_weight_khwc(w.permute(0, 2, 3, 1)); cin = Cin
    OH, OW = ref.shape[2], ref.shape[3]
    out = torch.empty(B, OH, OW, Cout, device=dev)
    xin, wk, bias_d = xin.to(dev), wk.to(dev), bias.to(dev)
    res_d = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    check(L.ssg_conv2d_nhwc_f32(ptr(xin), ptr(wk), ptr(bias_d), ptr(res_d), ptr(out), B, H, W, cin, Cout, k, k, stride, pad, int(relu), stream()), "conv")
    got = out.cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err     # fp32 accumulation over K <= 1152 terms


@pytest.mark.parametrize("cfg", [
    (3, 16, 8, 64, 64, 1, 1, 0, False, True),
    (2, 16, 8, 64, 256, 1, 1, 0, True, True),
    (2, 16, 8, 128, 128, 3, 1, 1, False, True),
    (2, 16, 8, 128, 128, 3, 2, 1, False, True),
    (2, 16, 8, 256, 512, 1, 2, 0, False, False),
    (5, 9, 7, 64, 64, 3, 1, 1, False, True),
    (2, 8, 4, 512, 512, 3, 1, 1, True, True),       # K = 4608
])
def test_conv_split_half_vs_fp64(cfg, L, dev):
    """Split-half convolution (fp16 matrix cores, hi/lo operands, fp32 accumulate) against an fp64 convolution,
    side by side with the pure fp32-MFMA kernel on the same data: the split path must be fp32-class --
    within 2x of the fp32 kernel's own error (+ the 2^-22 operand representation) and inside the same
    absolute bound the fp32 test uses."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _weight_scale
    B, H, W, Cin, Cout, k, stride, pad, use_res, relu = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000 + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0, :, 0, 0] *= 1e-4                      # a pixel of tiny activations (half-subnormal lo parts)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride, pad)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    from ssg_amd.resnet import pack_weight_khwc
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev); wk = pack_weight_khwc(w.permute(0, 2, 3, 1))
    OH, OW = ref.shape[2], ref.shape[3]
    bias_d = bias.to(dev)
    res_d = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    # fp32 MFMA kernel
    out32 = torch.empty(B, OH, OW, Cout, device=dev)
    check(L.ssg_conv2d_nhwc_f32(ptr(xin), ptr(wk.to(dev)), ptr(bias_d), ptr(res_d), ptr(out32), B, H, W, Cin, Cout, k, k, stride, pad, int(relu), stream()), "conv")
    # split-half kernel: encode activations / residual on the device, weights on the host
    sc = _weight_scale(wk)
    ws = _h8l8(wk * sc).to(dev)
    xs = torch.empty_like(xin); check(L.ssg_h8l8_encode(ptr(xin), ptr(xs), xin.numel(), 1.0, stream()), "enc")
    rs = None
    if use_res:
        rs = torch.empty_like(res_d); check(L.ssg_h8l8_encode(ptr(res_d), ptr(rs), res_d.numel(), 1.0, stream()), "enc")
    outs = torch.empty(B, OH, OW, Cout, device=dev)
    check(L.ssg_conv2d_nhwc_x(ptr(xs), ptr(ws), ptr(bias_d), ptr(rs), ptr(outs), B, H, W, Cin, Cout, k, k, stride, pad, int(relu), 3, 1.0 / sc, None, None, stream()), "convx")
    dec = torch.empty_like(outs); check(L.ssg_h8l8_decode(ptr(outs), ptr(dec), outs.numel(), 1.0, stream()), "dec")
    # fp32 output of the same split GEMM (flags = IN only)
    outp = torch.empty(B, OH, OW, Cout, device=dev)
    check(L.ssg_conv2d_nhwc_x(ptr(xs), ptr(ws), ptr(bias_d), ptr(res_d), ptr(outp), B, H, W, Cin, Cout, k, k, stride, pad, int(relu), 1, 1.0 / sc, None, None, stream()), "convx")
    scale = max(1.0, ref.abs().max().item())
    e32 = (out32.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    esp = (dec.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    print("conv %r: fp32-mfma err %.3g  split err %.3g" % (cfg, e32, esp))
    assert esp < 2e-5 * scale, (esp, e32)
    assert esp < 2.0 * e32 + 2.0 ** -21 * scale, (esp, e32)
    # the stored format is a fixed point: decode -> encode -> decode gives the same values back
    re = torch.empty_like(dec); check(L.ssg_h8l8_encode(ptr(dec), ptr(re), dec.numel(), 1.0, stream()), "enc")
    dec2 = torch.empty_like(dec); check(L.ssg_h8l8_decode(ptr(re), ptr(dec2), dec.numel(), 1.0, stream()), "dec")
    assert torch.equal(dec2, dec)
    # fp32 output of the same GEMM (flags = IN only) agrees with the decoded split output to the format's 2^-22
    assert (outp - dec).abs().max().item() <= 2.0 ** -21 * scale


@pytest.mark.parametrize("cfg", [
    (400, 16, 8, 256, 256, 3, 1, 1, False),        # layer3 3x3: 256 x 256 tiles (four LDS stages), K = 2304: 144 k-tiles
    (400, 16, 8, 1024, 256, 1, 1, 0, False),       # layer3 conv1: K = 1024 streamed from HBM
    (420, 16, 8, 96, 256, 3, 1, 1, False),         # K = 864: 54 k-tiles, two left over after the four-stage rounds; ragged last tile (M = 53 760)
    (400, 16, 8, 256, 1024, 1, 1, 0, True),        # conv3 + residual: 128 x 256 tiles (three stages) at this size
    # round 4: the layer2 and layer4 shapes of the unfused path as well
    (400, 32, 16, 512, 128, 1, 1, 0, False),       # layer2 conv1
    (400, 32, 16, 128, 128, 3, 1, 1, False),       # layer2 3x3
    (400, 32, 16, 128, 512, 1, 1, 0, True),        # layer2 conv3 + residual
    (300, 64, 32, 128, 128, 3, 2, 1, False),       # layer2 first block: 3x3 stride 2
    (400, 32, 16, 256, 256, 3, 2, 1, False),       # layer3 first block: 3x3 stride 2
    (400, 8, 4, 2048, 512, 1, 1, 0, False),        # layer4 conv1
    (400, 8, 4, 512, 512, 3, 1, 1, False),         # layer4 3x3
    (400, 8, 4, 512, 2048, 1, 1, 0, True),         # layer4 conv3 + residual
    (400, 16, 8, 512, 512, 3, 2, 1, False),        # layer4 first block: 3x3 stride 2
])
def test_conv_tile_shapes_agree_bitwise(cfg, L, dev):
    """The tile shape is a launch-time choice (conv.hip launch_conv_wide: 256 x 256 tiles with four LDS stages when at least 200
    of them exist, 128 x 256 / 128 x 128 tiles with three stages below that); every output element's reduction order is the same
    in all of them, so a large batch (tall tiles) must equal the same images run in small batches (short tiles) bit for bit.  The
    parity tests against the reference model use a handful of images and never reach the tall tiles."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _weight_scale, pack_weight_khwc
    B, H, W, Cin, Cout, k, stride, pad, use_res = cfg
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    wk = pack_weight_khwc(w.permute(0, 2, 3, 1))
    sc = _weight_scale(wk)
    ws = _h8l8(wk * sc).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
    OH = (H + 2 * pad - k) // stride + 1; OW = (W + 2 * pad - k) // stride + 1
    rs = None
    if use_res:
        r = torch.randn(B, OH, OW, Cout, generator=g).to(dev)
        rs = torch.empty_like(r); check(L.ssg_h8l8_encode(ptr(r), ptr(rs), r.numel(), 1.0, stream()), "enc")

    def run(lo, hi):
        out = torch.empty(hi - lo, OH, OW, Cout, device=dev)
        check(L.ssg_conv2d_nhwc_x(ptr(xs[lo:hi]), ptr(ws), ptr(bias), ptr(rs[lo:hi]) if use_res else None, ptr(out), hi - lo, H, W, Cin, Cout, k, k, stride, pad, 1, 3,
                                  1.0 / sc, None, None, stream()), "convx")
        return out
    big = run(0, B)
    small = torch.cat([run(lo, min(lo + 64, B)) for lo in range(0, B, 64)], 0)
    assert torch.equal(big.view(torch.int32), small.view(torch.int32))
    # and the values are right: fp64 convolution of a few images
    dec = torch.empty_like(big[:2]); check(L.ssg_h8l8_decode(ptr(big[:2].contiguous()), ptr(dec), dec.numel(), 1.0, stream()), "dec")
    ref = torch.nn.functional.conv2d(x[:2].cpu().permute(0, 3, 1, 2).double(), w.double(), bias.cpu().double(), stride, pad)
    if use_res:
        ref = ref + r[:2].cpu().permute(0, 3, 1, 2).double()
    ref = torch.relu(ref)
    assert (dec.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [
    (400, 16, 8, 256, 1024, 1, 1, 0, True),        # layer3 conv3 + residual: 128 x 256 tiles
    (401, 8, 4, 512, 2048, 1, 1, 0, True),         # layer4 conv3 + residual, ragged last tile (M = 12 832 = 100.25 x 128)
    (403, 8, 4, 2048, 512, 1, 1, 0, False),        # layer4 conv1: 256 x 256 tiles, ragged (M = 12 896 = 50.4 x 256)
    (400, 16, 8, 256, 256, 3, 1, 1, False),        # layer3 3x3: 256 x 256 tiles
    (400, 32, 16, 128, 512, 1, 1, 0, True),        # layer2 conv3 + residual
])
def test_conv_fast_epilogue_matches_the_general_one(cfg, L, dev):
    """Round 6: the LDS-DMA kernels finish the embedding's own launches (split-half in / out, ReLU, per-channel scales) through a
    straight-line epilogue with compile-time switches, buffer-resource addressing and the residual pieces of the next patch in flight;
    a large batch (that epilogue) must equal the same images in small batches (register-staged kernel, general epilogue) bit for bit,
    twice in a row, ragged last tiles included; an activation beyond the half range must raise the range flag on either path."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _row_scales, pack_weight_khwc
    B, H, W, Cin, Cout, k, stride, pad, use_res = cfg
    g = torch.Generator().manual_seed(7 * B + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5 * (10.0 ** torch.randint(-2, 2, (Cout, 1, 1, 1), generator=g).float())
    wk = pack_weight_khwc(w.permute(0, 2, 3, 1))
    sc = _row_scales(wk)
    ws = _h8l8(wk * sc.view(-1, 1)).to(dev)
    cs = (1.0 / sc).contiguous().to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
    OH = (H + 2 * pad - k) // stride + 1; OW = (W + 2 * pad - k) // stride + 1
    rs = None
    if use_res:
        r = torch.randn(B, OH, OW, Cout, generator=g).to(dev)
        rs = torch.empty_like(r); check(L.ssg_h8l8_encode(ptr(r), ptr(rs), r.numel(), 1.0, stream()), "enc")
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(lo, hi, b=bias):
        out = torch.empty(hi - lo, OH, OW, Cout, device=dev)
        check(L.ssg_conv2d_nhwc_x(ptr(xs[lo:hi]), ptr(ws), ptr(b), ptr(rs[lo:hi]) if use_res else None, ptr(out), hi - lo, H, W, Cin, Cout, k, k, stride, pad, 1, 3,
                                  1.0, ptr(cs), ptr(flag), stream()), "convx")
        return out
    big = run(0, B)
    small = torch.cat([run(lo, min(lo + 8, B)) for lo in range(0, B, 8)], 0)
    assert torch.equal(big.view(torch.int32), small.view(torch.int32))
    assert torch.equal(run(0, B).view(torch.int32), big.view(torch.int32))
    assert int(flag.item()) == 0
    dec = torch.empty_like(big[:2]); check(L.ssg_h8l8_decode(ptr(big[:2].contiguous()), ptr(dec), dec.numel(), 1.0, stream()), "dec")
    ref = torch.nn.functional.conv2d(x[:2].cpu().permute(0, 3, 1, 2).double(), w.double(), bias.cpu().double(), stride, pad)
    if use_res:
        ref = ref + r[:2].cpu().permute(0, 3, 1, 2).double()
    ref = torch.relu(ref)
    assert (dec.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # one channel pushed out of the half range: the flag goes up (large batch = this epilogue)
    hot = bias.clone(); hot[Cout // 2 + 3] = 1.0e5
    run(0, B, hot)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("M", [51200, 25664, 128 * 257 + 37])
def test_conv_pair_matches_the_two_launches(M, L, dev):
    """Round 6 (experimental kernel, SSG_CONV_PAIR=1): conv3 + residual of a layer3 identity block and conv1 of the next block as ONE launch
    (ssg_conv_pair_nhwc_x: `out` walked in 128-channel chunks, the chunk re-encoded into an LDS stash that is the second GEMM's pixel operand)
    must equal the two ssg_conv2d_nhwc_x launches bit for bit -- both tensors, twice in a row, ragged last tiles included -- and raise the range
    flag for an activation beyond the half range."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _row_scales
    K1, C, N2 = 256, 1024, 256
    assert L.ssg_conv_pair_supported(K1, C, N2)
    g = torch.Generator().manual_seed(M)
    y2 = torch.relu(torch.randn(M, K1, generator=g)).to(dev)
    res = torch.relu(torch.randn(M, C, generator=g)).to(dev)
    w3 = torch.randn(C, K1, generator=g) * (2.0 / K1) ** 0.5 * (10.0 ** torch.randint(-2, 2, (C, 1), generator=g).float())
    w1 = torch.randn(N2, C, generator=g) * (2.0 / C) ** 0.5 * (10.0 ** torch.randint(-2, 2, (N2, 1), generator=g).float())
    s3, s1 = _row_scales(w3), _row_scales(w1)
    w3s, w1s = _h8l8(w3 * s3.view(-1, 1)).to(dev), _h8l8(w1 * s1.view(-1, 1)).to(dev)
    cs3, cs1 = (1.0 / s3).contiguous().to(dev), (1.0 / s1).contiguous().to(dev)
    b3, b1 = torch.randn(C, generator=g).to(dev), torch.randn(N2, generator=g).to(dev)
    y2s = torch.empty_like(y2); check(L.ssg_h8l8_encode(ptr(y2), ptr(y2s), y2.numel(), 1.0, stream()), "enc")
    rs = torch.empty_like(res); check(L.ssg_h8l8_encode(ptr(res), ptr(rs), res.numel(), 1.0, stream()), "enc")
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def two(bias3):
        out = torch.empty(M, C, device=dev); y1 = torch.empty(M, N2, device=dev)
        check(L.ssg_conv2d_nhwc_x(ptr(y2s), ptr(w3s), ptr(bias3), ptr(rs), ptr(out), M, 1, 1, K1, C, 1, 1, 1, 0, 1, 3, 1.0, ptr(cs3), ptr(flag), stream()), "conv3")
        check(L.ssg_conv2d_nhwc_x(ptr(out), ptr(w1s), ptr(b1), None, ptr(y1), M, 1, 1, C, N2, 1, 1, 1, 0, 1, 3, 1.0, ptr(cs1), ptr(flag), stream()), "conv1")
        return out, y1

    def one(bias3):
        out = torch.empty(M, C, device=dev); y1 = torch.empty(M, N2, device=dev)
        check(L.ssg_conv_pair_nhwc_x(ptr(y2s), ptr(w3s), ptr(bias3), ptr(cs3), ptr(rs), ptr(out), ptr(w1s), ptr(b1), ptr(cs1), ptr(y1), M, K1, C, N2, ptr(flag), stream()), "pair")
        return out, y1
    o2, y2l = two(b3)
    for _ in range(2):
        o1, y1l = one(b3)
        assert torch.equal(o1.view(torch.int32), o2.view(torch.int32))
        assert torch.equal(y1l.view(torch.int32), y2l.view(torch.int32))
    assert int(flag.item()) == 0
    # values: float64 on a few rows
    dec = torch.empty(4, N2, device=dev); check(L.ssg_h8l8_decode(ptr(y1l[:4].contiguous()), ptr(dec), dec.numel(), 1.0, stream()), "dec")
    o64 = torch.relu(y2[:4].cpu().double() @ w3.double().t() + b3.cpu().double() + res[:4].cpu().double())
    r64 = torch.relu(o64 @ w1.double().t() + b1.cpu().double())
    assert (dec.cpu().double() - r64).abs().max().item() < 3e-5 * max(1.0, r64.abs().max().item())
    hot = b3.clone(); hot[C // 2 + 5] = 1.0e5
    one(hot)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("cfg", [
    (400, 32, 16, 128, 64, 32, 256, 2, 512),       # layer2 first block: conv3 (128 ch) | downsample (256 ch at stride 2) -> 512
    (400, 16, 8, 256, 32, 16, 512, 2, 1024),       # layer3 first block
    (400, 8, 4, 512, 16, 8, 1024, 2, 2048),        # layer4 first block
    (200, 64, 32, 64, 64, 32, 64, 1, 256),         # layer1 first block (stride-1 downsample), the unfused fallback
])
def test_conv_dual_tile_shapes_agree_bitwise(cfg, L, dev):
    """the DUAL instantiation (conv3 | downsample as one GEMM over the concatenated K, ssg_conv1x1_dual_nhwc_x) --
    a large batch (conv_dma_kernel<256, false, 128, true>, 128 x 256 tiles) must equal the same images in small batches (the
    register-staged kernel on short tiles) bit for bit, twice in a row (the LDS-DMA race of round 3 showed up run to run), and a
    float64 convolution of two images."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _row_scales
    B, H, W, Cin, H2, W2, Cin2, s2, Cout = cfg
    g = torch.Generator().manual_seed(B + Cin + Cout)
    o = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x = torch.randn(B, H2, W2, Cin2, generator=g).to(dev)
    w3 = torch.randn(Cout, Cin, generator=g) * (1.0 / Cin) ** 0.5
    wd = torch.randn(Cout, Cin2, generator=g) * (1.0 / Cin2) ** 0.5
    wcat = torch.cat([w3, wd], 1)
    sc = _row_scales(wcat)
    ws = _h8l8(wcat * sc.view(-1, 1)).to(dev)
    cs = (1.0 / sc).contiguous().to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    os_ = torch.empty_like(o); check(L.ssg_h8l8_encode(ptr(o), ptr(os_), o.numel(), 1.0, stream()), "enc")
    xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")

    def run(lo, hi):
        out = torch.empty(hi - lo, H, W, Cout, device=dev)
        check(L.ssg_conv1x1_dual_nhwc_x(ptr(os_[lo:hi]), ptr(xs[lo:hi]), ptr(ws), ptr(bias), ptr(out), hi - lo, H, W, Cin, H2, W2, Cin2, s2, Cout, 1, 3,
                                        1.0, ptr(cs), None, stream()), "dual")
        return out
    big = run(0, B)
    small = torch.cat([run(lo, min(lo + 40, B)) for lo in range(0, B, 40)], 0)
    assert torch.equal(big.view(torch.int32), small.view(torch.int32))
    assert torch.equal(run(0, B).view(torch.int32), big.view(torch.int32))
    dec = torch.empty_like(big[:2]); check(L.ssg_h8l8_decode(ptr(big[:2].contiguous()), ptr(dec), dec.numel(), 1.0, stream()), "dec")
    ref = torch.einsum("bhwc,oc->bhwo", o[:2].cpu().double(), w3.double()) + torch.einsum("bhwc,oc->bhwo", x[:2, ::s2, ::s2][:, :H, :W].cpu().double(), wd.double())
    ref = torch.relu(ref + bias.cpu().double())
    assert (dec.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_embedding_vs_reference_golden(golden, dev, precision):
    """HIP ResNet-50 embed (orig + flip, L2 norm) vs the real reference model's features.
    Tolerance 5e-6 absolute on unit-norm 2048-d features (fp32 conv accumulation order)."""
    import ssg_amd
    g = golden("embed_ref.npz")
    imgs = torch.randn(4, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    for S in (2, 1):
        m = ssg_amd.create("resnet50", num_classes=0, num_split=S, cluster=False, seed=int(g["weight_seed"]), precision=precision).cuda().eval()
        ref = g["feats_S%d" % S]
        got = m.embed_with_flip(imgs)
        print("embedding S=%d precision=%s: max |err| vs reference %.3g" % (S, precision, np.abs((got.cpu().numpy() if got.dim() == 3 else got.cpu().numpy()[None]) - ref).max()))
        got = got.cpu().numpy() if got.dim() == 3 else got.cpu().numpy()[None]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 5e-6, np.abs(got - ref).max()
        # plain forward surface: model(x, for_eval)[0] (reid/feature_extraction/cnn.py:16)
        x1, x2 = m(imgs[:2], False)
        if S > 1:
            assert isinstance(x1, list) and len(x1) == S + 1 and x1[0].shape == (2, 2048) and x2.shape == (2, 2048)
            cat, _ = m(imgs[:2], True)
            assert cat.shape == (2, (S + 1) * 2048)


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_embedding_vs_wide_reference_goldens(golden, dev, precision):
    """round 5: 16 images under the seeded Kaiming weights, and 8 images under checkpoint-like BatchNorm statistics
    (folded per-channel scales spanning > 10^4) -- both fixtures written by the REAL reference model (reid.models.create +
    reid.evaluators.extract_features, tools/make_golden.py --only-embed-wide).  The split-half default and the fp32-MFMA path must
    both stay inside 5e-6 absolute on the unit-norm features; no overflow fallback may be needed."""
    import warnings
    import ssg_amd
    from synth import checkpoint_like_state_dict
    for fname, mk in (("embed_ref16.npz", lambda s: ssg_amd.synthetic_state_dict(seed=s)), ("embed_ckpt_ref.npz", checkpoint_like_state_dict)):
        g = golden(fname)
        n = int(g["n"])
        imgs = torch.randn(n, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
        m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, pretrained=False, precision=precision).cuda().eval()
        m.load_state_dict(mk(int(g["weight_seed"])), strict=False)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            got = m.embed_with_flip(imgs).cpu().numpy()
        ref = g["feats_S2"]
        assert got.shape == ref.shape
        err = float(np.abs(got - ref).max())
        print("embedding %s precision=%s: max |err| vs the reference model %.3g" % (fname, precision, err))
        assert err < 5e-6, (fname, err)
        # the dictionary surface on uneven batches gives the same rows (reid/evaluators.py:18-60)
        if precision == "split":
            names = ["f%d" % i for i in range(n)]
            loader = [(imgs[:5], names[:5], list(range(5)), [0] * 5), (imgs[5:], names[5:], list(range(5, n)), [0] * (n - 5))]
            feats, _ = ssg_amd.extract_features(m, loader, print_freq=0, for_eval=False)
            assert all(np.array_equal(feats[f][s].numpy(), got[s, i]) for i, f in enumerate(names) for s in range(3))


def test_embed_with_flip_on_two_streams_equals_one_stream(dev):
    """`embed_with_flip` runs the original and the flipped forward on two HIP streams (resnet.py, flip_streams): the result must be
    bit-identical to the one-stream order, also over back-to-back calls of different batches and sizes (tensors allocated on a side
    stream's pool are consumed on the caller's stream, then recycled by the next call) and on a non-default caller stream."""
    import ssg_amd
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=2, pretrained=False).cuda().eval()
    g = torch.Generator().manual_seed(7)
    batches = [torch.randn(b, 3, 256, 128, generator=g).cuda() for b in (5, 16, 3, 16, 9, 1)]
    m.flip_streams = False
    want = [m.embed_with_flip(x).clone() for x in batches]
    m.flip_streams = True
    got = [m.embed_with_flip(x) for x in batches]              # no synchronisation in between
    caller = torch.cuda.Stream()
    caller.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(caller):
        got2 = [m.embed_with_flip(x * 1.0) for x in batches]   # inputs produced on the caller's stream right before the call
    torch.cuda.synchronize()
    for w, a, b in zip(want, got, got2):
        assert torch.equal(w, a) and torch.equal(w, b)
    assert not m._overflowed()


def test_extract_features_dropin(golden, dev):
    """reid/evaluators.py:18 call surface: dict fname -> list of S+1 CPU vectors, dict fname -> pid."""
    import ssg_amd
    g = golden("embed_ref.npz")
    imgs = torch.randn(4, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=int(g["weight_seed"])).cuda()
    loader = [(imgs[:3], ["a", "b", "c"], [7, 8, 9], [0, 0, 0]), (imgs[3:], ["d"], [5], [1])]
    feats, labels = ssg_amd.extract_features(m, loader, for_eval=False)
    assert list(feats.keys()) == ["a", "b", "c", "d"] and labels["d"] == 5
    assert isinstance(feats["a"], list) and len(feats["a"]) == 3 and feats["a"][0].device.type == "cpu"
    ref = g["feats_S2"]
    for i, f in enumerate("abcd"):
        for s in range(3):
            assert np.abs(feats[f][s].numpy() - ref[s, i]).max() < 5e-6
    feats_e, _ = ssg_amd.extract_features(m, loader, for_eval=True)
    assert feats_e["a"].shape == (3 * 2048,) and abs(float(feats_e["a"].norm()) - 1.0) < 1e-5


# ------------------------------------------------------------------ edge regimes of the sparse kernels
def test_wide_neighbourhoods_vs_oracle(dev, ora):
    """k1=60, k2=16 on weakly clustered data: V rows near the 64-lane limit, inverted lists longer
    than one wave, touched-column list overflow in the Jaccard kernel (dense epilogue path)."""
    from ssg_amd import rerank, cluster
    N, d = 4500, 48
    rng = np.random.default_rng(12)
    tgt = rng.standard_normal((N, d)).astype(np.float32); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    src = rng.standard_normal((700, d)).astype(np.float32); src /= np.linalg.norm(src, axis=1, keepdims=True)
    st = {}
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k1=60, k2=16, lambda_value=0.2, stages=st)
    oe, of, ost = ora.re_ranking(src, tgt, k1=60, k2=16, lambda_value=0.2, stages=True)
    assert np.array_equal(st["rank"].cpu().numpy(), ost["rank"])
    assert np.array_equal(_sparse_to_dense(st["v_idx"], st["v_val"], st["v_nnz"], N), bits(ost["V"])), "V"
    assert np.array_equal(_sparse_to_dense(st["q_idx"], st["q_val"], st["q_nnz"], N), bits(ost["V_qe"])), "V_qe"
    got, ref = bits(st["Jp"].cpu().numpy()), bits(ost["jaccard_scaled"])
    touched = (ost["jaccard"] != np.float16(1.0)).sum(axis=1)
    assert touched.max() > 3072, "test should overflow the touched list (the sparse patch path is covered by the k1=20 tests)"
    bad = np.argwhere(got != ref)
    assert len(bad) == 0, (len(bad), bad[:5].tolist(), [(hex(got[a, b]), hex(ref[a, b])) for a, b in bad[:5]], st["q_nnz"].cpu().numpy()[bad[:5, 0]].tolist())
    assert np.array_equal(h.final_dist().cpu().numpy(), of)
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    assert (eps, cnt, top) == ora.eps_rule(of, 1.6e-3)
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), ora.dbscan(of, eps, 4))


def test_more_than_one_column_chunk_vs_oracle(dev, ora):
    """N > 32768: the Jaccard accumulator row is processed in two LDS column chunks."""
    from ssg_amd import rerank
    N, d = 33000, 32
    tgt = clustered(N, d, 5); src = clustered(2000, d, 6, intra=0.7)
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=0.1)
    oe, of, ost = ora.re_ranking(src, tgt, lambda_value=0.1, stages=True)
    assert np.array_equal(bits(h.euclid.cpu().numpy()), bits(oe))
    assert np.array_equal(bits(h.M.cpu().numpy()), bits(ost["jaccard_scaled"]))


def _check_sparse_copy(h, N):
    """the sparse copy S against the dense J' it was written with: every packed entry equals J' at its (row, column), no column twice,
    and every column that is not in S holds the constant J'(0)"""
    sp = h.sparse
    assert sp is not None and int(sp["cursor"][1].item()) == 0
    Jp = bits(h.M.cpu().numpy())
    pool = sp["pool"].cpu().numpy().view(np.uint32)
    off = sp["seg_off"].cpu().numpy().reshape(h.nrows, sp["nseg"]); ln = sp["seg_len"].cpu().numpy().reshape(h.nrows, sp["nseg"])
    assert int(ln.sum()) == int(sp["cursor"][0].item())
    jp0 = np.uint16(sp["jp0"])
    rng = np.random.default_rng(1)
    rows = np.unique(np.concatenate([np.arange(min(8, h.nrows)), rng.integers(0, h.nrows, 40), [h.nrows - 1]]))
    for il in rows:
        seen = np.zeros(N, bool)
        for sg in range(sp["nseg"]):
            e = pool[off[il, sg]:off[il, sg] + ln[il, sg]]
            col = (e & 0x1FFFF).astype(np.int64); val = (e >> 17).astype(np.uint16)
            assert (col // 32768 == sg).all() and not seen[col].any() and len(np.unique(col)) == len(col)
            assert np.array_equal(Jp[il, col], val), (il, sg)
            seen[col] = True
        assert (Jp[il, ~seen] == jp0).all(), il


def test_jaccard_second_generation_and_sparse_copy(dev, ora, monkeypatch):
    """round 4: ssg_jaccard_rows2 (every J' line written once from the LDS row, touched columns marked with bit 15) against the first
    generation kernel bit for bit -- rows that start off 16-byte boundaries (N % 8 != 0), the touched-list overflow path (k1 = 60), two
    column chunks (N > 32768) -- and the sparse copy S it emits against the dense rows; then the eps rule and DBSCAN through S
    against the dense passes (SSG_SPARSE=0) incl. the cases where S must NOT be used: eps above J'(0), a threshold above J'(0)
    (rho = 0.5: decided on the device), an overflowed pool."""
    from ssg_amd import rerank, cluster
    rng = np.random.default_rng(12)
    cases = []
    t = clustered(1531, 64, 7); cases.append(("ragged", clustered(400, 64, 8, intra=0.7), t, dict(lambda_value=0.3)))
    t = rng.standard_normal((4500, 48)).astype(np.float32); t /= np.linalg.norm(t, axis=1, keepdims=True)
    sr = rng.standard_normal((700, 48)).astype(np.float32); sr /= np.linalg.norm(sr, axis=1, keepdims=True)
    cases.append(("wide", sr, t, dict(k1=60, k2=16, lambda_value=0.2)))
    cases.append(("chunks", clustered(2000, 32, 6, intra=0.7), clustered(33000, 32, 5), dict(lambda_value=0.1)))
    cases.append(("hard", hard_clustered(2000, 128, 12, intra=0.7), hard_clustered(6000, 128, 11), dict(lambda_value=0.3)))
    for name, src, tgt, kw in cases:
        N = tgt.shape[0]
        s_d, t_d = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
        monkeypatch.setenv("SSG_SPARSE", "0")
        h1 = rerank.re_ranking_device(s_d, t_d, **kw)
        assert h1.sparse is None
        e1 = cluster.eps_rule(h1, 1.6e-3)
        l1 = cluster.DBSCAN(eps=e1[0], min_samples=4, metric="precomputed").fit_predict(h1)
        big = 0.95 * (1.0 - kw["lambda_value"]) + 0.4          # an eps above J'(0): dense region query on both handles
        l1big = cluster.DBSCAN(eps=big, min_samples=4, metric="precomputed").fit_predict(h1) if N <= 6000 else None
        e1half = cluster.eps_rule(h1, 0.5) if N <= 6000 else None
        monkeypatch.delenv("SSG_SPARSE")
        if name == "wide":
            # rows with more than 3072 touched columns (the LDS list overflows: dense patch pass): the default pool (1024 entries per row)
            # cannot hold them -- S is marked unusable and the consumers go dense; a pool of N entries per row holds everything
            h0 = rerank.re_ranking_device(s_d, t_d, **kw)
            assert torch.equal(h1.M.view(torch.int16), h0.M.view(torch.int16)) and not h0.sparse_complete()
            assert cluster.eps_rule(h0, 1.6e-3) == e1
            assert np.array_equal(cluster.DBSCAN(eps=e1[0], min_samples=4, metric="precomputed").fit_predict(h0), l1)
            monkeypatch.setenv("SSG_SPARSE_ROW_ENTRIES", str(N))
        h2 = rerank.re_ranking_device(s_d, t_d, **kw)
        monkeypatch.delenv("SSG_SPARSE_ROW_ENTRIES", raising=False)
        assert torch.equal(h1.M.view(torch.int16), h2.M.view(torch.int16)), name
        assert h2.sparse_complete(), name
        _check_sparse_copy(h2, N)
        assert cluster.eps_rule(h2, 1.6e-3) == e1, name
        assert np.array_equal(cluster.DBSCAN(eps=e1[0], min_samples=4, metric="precomputed").fit_predict(h2), l1), name
        if N <= 6000:
            assert float(np.uint16(h2.sparse["jp0"]).view(np.float16)) < big
            assert np.array_equal(cluster.DBSCAN(eps=big, min_samples=4, metric="precomputed").fit_predict(h2), l1big), name
            assert cluster.eps_rule(h2, 0.5) == e1half, name
        # a pool that is too small: S is marked unusable, everything falls back to the dense passes
        monkeypatch.setenv("SSG_SPARSE_ROW_ENTRIES", "8")
        h3 = rerank.re_ranking_device(s_d, t_d, **kw)
        monkeypatch.delenv("SSG_SPARSE_ROW_ENTRIES")
        assert torch.equal(h1.M.view(torch.int16), h3.M.view(torch.int16)) and not h3.sparse_complete()
        assert cluster.eps_rule(h3, 1.6e-3) == e1
        assert np.array_equal(cluster.DBSCAN(eps=e1[0], min_samples=4, metric="precomputed").fit_predict(h3), l1)
        del h1, h2, h3


def test_sparse_passes_mixed_rows_equal_dense(dev):
    """round 4: the per-row decision of the sparse passes.  With a bound in the MIDDLE of the rows' floors J'(0) + lambda * half(v_i + min v)
    about half of the rows are walked through the sparse copy S and the others are flagged for the dense pass queued behind: the eps
    compaction (same key multiset, same zero count) and the region query (same neighbour counts, same edge set) must equal the dense
    passes alone, for a bound below every floor (all rows sparse), in the middle (mixed) and above every floor (all rows dense)."""
    from ssg_amd import rerank, _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib(); st = stream()
    tgt = hard_clustered(3000, 96, 21); src = hard_clustered(800, 96, 22, intra=0.7)
    lam = 0.3
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=lam)
    sp = h.sparse
    assert sp is not None and h.sparse_complete()
    N = h.N
    v = h.v.cpu().numpy(); jp0 = np.uint16(sp["jp0"]).view(np.float16)
    floors = jp0.astype(np.float64) + (v + v.min()).astype(np.float16).astype(np.float64) * lam        # f64(J'(0)) + f64(half(v_i + vmin)) * lambda
    assert floors.max() > floors.min()
    for name, bound in (("all sparse", float(floors.min()) * 0.98), ("mixed", float(np.median(floors))), ("all dense", float(floors.max()) * 1.02)):
        # ---- eps compaction with a hand-made threshold
        thr3 = torch.zeros(5, dtype=torch.int64, device=dev); thr3[0] = int(np.float32(bound).view(np.uint32))
        cap = 1 << 23
        outs = []
        for sparse in (True, False):
            buf = torch.empty(cap, dtype=torch.int64, device=dev); cur = torch.zeros(3, dtype=torch.int64, device=dev)
            if sparse:
                check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), N, 0, N, lam, ptr(thr3), ptr(buf), cap, ptr(cur), ptr(sp["pool"]), ptr(sp["seg_off"]),
                                                ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), st), "s")
                ndense = int(sp["rowmask"].sum().item())
            else:
                check(L.ssg_eps_compact_below(ptr(h.M), ptr(h.v), N, 0, N, 0, lam, ptr(thr3), ptr(buf), cap, ptr(cur), st), "d")
            got, zeros = int(cur[0].item()), int(cur[1].item())
            assert got <= cap
            outs.append((np.sort(buf[:got].cpu().numpy()), zeros))
        assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1], name
        assert (ndense == 0) if name == "all sparse" else (ndense == N) if name == "all dense" else (0 < ndense < N), (name, ndense)
        # ---- region query with eps = the same bound
        res = []
        for sparse in (True, False):
            ecap = 1 << 23
            cnt = torch.empty(N, dtype=torch.int32, device=dev); edges = torch.empty((ecap, 2), dtype=torch.int32, device=dev); cur = torch.zeros(2, dtype=torch.int64, device=dev)
            if sparse:
                check(L.ssg_region_query_s(ptr(h.M), ptr(h.v), N, 0, N, lam, bound, ptr(sp["pool"]), ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]),
                                           ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), ptr(cnt), ptr(edges), ecap, ptr(cur), st), "s")
            else:
                check(L.ssg_region_query(ptr(h.M), ptr(h.v), N, 0, N, 0, lam, bound, ptr(cnt), ptr(edges), ecap, ptr(cur), st), "d")
            ne = int(cur[0].item())
            assert ne <= ecap
            e = edges[:ne].cpu().numpy().astype(np.int64)
            res.append((cnt.cpu().numpy(), np.sort(e[:, 0] * N + e[:, 1])))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), name


def test_pairwise_distance_dropin(dev):
    """reid/evaluators.py:63-85 (float32): both branches vs the torch CPU formula; 2e-5 absolute
    on distances of unit-norm features (fp32 GEMM accumulation order)."""
    from collections import OrderedDict
    import ssg_amd
    g = torch.Generator().manual_seed(3)
    feats = OrderedDict()
    for i in range(70):
        f = torch.randn(200, generator=g); feats["f%d" % i] = f / f.norm()
    query = [("f%d" % i, 0, 0) for i in range(0, 30)]
    gallery = [("f%d" % i, 0, 0) for i in range(20, 70)]
    x = torch.stack([feats[f] for f, _, _ in query]); y = torch.stack([feats[f] for f, _, _ in gallery])
    ref = x.pow(2).sum(1, keepdim=True).expand(30, 50) + y.pow(2).sum(1, keepdim=True).expand(50, 30).t() - 2 * x @ y.t()
    got = ssg_amd.pairwise_distance(feats, query, gallery)
    assert got.shape == (30, 50) and got.device.type == "cpu" and (got - ref).abs().max() < 2e-5
    allx = torch.stack(list(feats.values()))
    ref_self = allx.pow(2).sum(1, keepdim=True) * 2 - 2 * allx @ allx.t()
    got_self = ssg_amd.pairwise_distance(feats)
    assert got_self.shape == (70, 70) and (got_self - ref_self).abs().max() < 2e-5


def test_pairwise_distance_vs_reference_golden(golden, dev):
    """a11 against the REFERENCE's own output (tests/golden/pairwise.npz, generated by tools/make_golden.py from
    reid/evaluators.py:63-85 as shipped, float32 torch on the CPU): query x gallery branch and features-only branch, unit-norm
    200-d features and un-normalised 2048-d ones.  Tolerance: float32 GEMM accumulation order, 2e-5 relative to the largest distance."""
    from collections import OrderedDict
    import ssg_amd
    g = golden("pairwise.npz")
    for tag in ("u", "r"):
        F = torch.from_numpy(g["feats_" + tag]); n = F.shape[0]; nq, g0 = int(g["nq_" + tag]), int(g["g0_" + tag])
        feats = OrderedDict(("f%03d" % i, F[i]) for i in range(n))
        query = [("f%03d" % i, 0, 0) for i in range(0, nq)]
        gallery = [("f%03d" % i, 0, 0) for i in range(g0, n)]
        qg = ssg_amd.pairwise_distance(feats, query, gallery).numpy()
        full = ssg_amd.pairwise_distance(feats).numpy()
        tol = 2e-5 * max(1.0, float(np.abs(g["self_" + tag]).max()))
        assert qg.shape == g["qg_" + tag].shape and np.abs(qg - g["qg_" + tag]).max() < tol, tag
        assert full.shape == g["self_" + tag].shape and np.abs(full - g["self_" + tag]).max() < tol, tag


def test_x2_branch_matches_torch(dev):
    """resnet.py:112-117 feat -> feat_bn -> relu on the pooled feature (HIP GEMM with folded BN)."""
    import ssg_amd
    m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=3).cuda().eval()
    sd = m.state_dict()
    g = torch.Generator().manual_seed(0)
    sd["feat_bn.running_mean"] = torch.randn(2048, generator=g) * 0.01; sd["feat_bn.running_var"] = torch.rand(2048, generator=g) + 0.5
    sd["feat_bn.weight"] = torch.rand(2048, generator=g) + 0.5; sd["feat_bn.bias"] = torch.randn(2048, generator=g) * 0.01
    m.load_state_dict(sd)
    x = torch.randn(3, 3, 256, 128, generator=g)
    x1, x2 = m(x, False)
    gap = x1.cpu().double()
    ref = gap @ sd["feat.weight"].double().t()
    ref = (ref - sd["feat_bn.running_mean"].double()) / torch.sqrt(sd["feat_bn.running_var"].double() + 1e-5) * sd["feat_bn.weight"].double() + sd["feat_bn.bias"].double()
    ref = torch.relu(ref)
    assert x2.shape == (3, 2048)
    assert (x2.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())     # fp32 GEMM over K=2048


def test_re_ranking_init_vs_reference_golden(golden, dev, ora):
    """float32 cosine variant (reid/rerank.py:171-234, rerank_initial.py:40-99) vs the reference's own
    output; tolerance 2e-5 (fp32 GEMM / exp are not bit reproducible across BLAS and libm)."""
    import ssg_amd
    g = golden("rerank_init.npz")
    for tag in ("a", "b"):
        q, gal = g["q_" + tag], g["g_" + tag]
        k1, k2, lam = int(g["k1_" + tag]), int(g["k2_" + tag]), float(g["lam_" + tag])
        out = ssg_amd.re_ranking_init(q, gal, k1=k1, k2=k2, lambda_value=lam)
        ref = g["final_" + tag]
        assert out.shape == ref.shape and out.dtype == np.float32
        assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()
        out2 = ssg_amd.re_ranking_init_dist(np.dot(q, gal.T), np.dot(q, q.T), np.dot(gal, gal.T), k1=k1, k2=k2, lambda_value=lam)
        assert np.abs(out2 - ref).max() < 2e-5
    # a larger ragged case against the oracle
    x = clustered(1000 + 333, 72, 9)
    out = ssg_amd.re_ranking_init(x[:333], x[333:], k1=20, k2=6, lambda_value=0.3)
    ref = ora.re_ranking_init(x[:333], x[333:], k1=20, k2=6, lambda_value=0.3)
    assert out.shape == (333, 1000) and np.abs(out - ref).max() < 2e-5


# ------------------------------------------------------------------ retrieval metrics
def test_rank_metrics_vs_reference_golden(golden, dev):
    """HIP cmc / mean_ap vs the reference's own outputs (tests/golden/eval_cases.npz).  CMC curves are exact
    (integer counts / count); AP is a float64 sum of at most |matches| terms in a different order than sklearn's
    pairwise np.sum -> 1e-12."""
    import io, contextlib
    from ssg_amd import ranking
    g = golden("eval_cases.npz")
    for tag in "abc":
        args = (g["dist_" + tag], g["qid_" + tag], g["gid_" + tag], g["qcam_" + tag], g["gcam_" + tag])
        first, ap = ranking.per_query(*args)
        assert np.array_equal(first.cpu().numpy(), g["first_" + tag])
        ref_ap = g["ap_" + tag]
        assert np.array_equal(np.isnan(ap.cpu().numpy()), np.isnan(ref_ap))
        assert np.nanmax(np.abs(ap.cpu().numpy() - ref_ap)) < 1e-12
        assert abs(ranking.mean_ap(*args) - float(g["map_" + tag])) < 1e-12
        assert np.array_equal(ranking.cmc(*args, first_match_break=True), g["cmc_" + tag])
        assert np.array_equal(ranking.cmc(*args), g["cmc_all_" + tag]), "all-shots protocol (ranking.py defaults) vs the reference"
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            top1 = ranking.evaluate_all(torch.from_numpy(args[0]).to(dev), query_ids=args[1], gallery_ids=args[2], query_cams=args[3], gallery_cams=args[4])
        assert top1 == g["cmc_" + tag][0] and "Mean AP" in buf.getvalue() and "top-1" in buf.getvalue()
    with pytest.raises(NotImplementedError):
        ranking.cmc(*args, single_gallery_shot=True)       # random gallery subsets from numpy's global RNG: not reproducible off the host
    with pytest.raises(RuntimeError):
        ranking.mean_ap(np.array([[0.1, 0.2, 0.3]], np.float32), [1], [1, 2, 3], [0], [0, 1, 1])


def test_rank_metrics_large_vs_oracle(dev):
    """Market-sized block (3368 x 15913 in the real set; 700 x 6000 here so the CPU oracle stays in seconds) with
    duplicated gallery rows (exact ties), a strided (padded) device distance block and the separate-camera protocol."""
    from oracle import eval_oracle
    from ssg_amd import ranking
    rng = np.random.default_rng(5)
    m, n, nid = 700, 6000, 150
    gid = rng.integers(0, nid, n).astype(np.int32); gcam = rng.integers(0, 6, n).astype(np.int32)
    qid = rng.integers(0, nid + 5, m).astype(np.int32); qcam = rng.integers(0, 6, m).astype(np.int32)
    c = rng.standard_normal((nid + 5, 24)); gf = (c[gid] + 1.3 * rng.standard_normal((n, 24))).astype(np.float32)
    qf = (c[qid] + 1.3 * rng.standard_normal((m, 24))).astype(np.float32)
    gf[3000:3100] = gf[:100]; gid[3000:3100] = gid[:100]
    dist = ((qf ** 2).sum(1)[:, None] + (gf ** 2).sum(1)[None, :] - 2 * qf @ gf.T).astype(np.float32)
    padded = torch.zeros(m, n + 64, device=dev); padded[:, :n] = torch.from_numpy(dist).to(dev)
    first, ap = ranking.per_query(padded[:, :n], qid, gid, qcam, gcam)
    ofirst, oap = eval_oracle.per_query(dist, qid, gid, qcam, gcam)
    assert np.array_equal(first.cpu().numpy(), ofirst)
    assert np.nanmax(np.abs(ap.cpu().numpy() - oap)) < 1e-12
    assert np.array_equal(ranking.cmc(dist, qid, gid, qcam, gcam, first_match_break=True, separate_camera_set=True),
                          eval_oracle.cmc(dist, qid, gid, qcam, gcam, first_match_break=True, separate_camera_set=True))
    for sep in (False, True):        # all-shots protocol incl. exact ties between duplicated gallery rows
        assert np.array_equal(ranking.cmc(dist, qid, gid, qcam, gcam, separate_camera_set=sep, topk=50),
                              eval_oracle.cmc(dist, qid, gid, qcam, gcam, separate_camera_set=sep, topk=50)), sep


def test_evaluator_dropin(golden, dev):
    """reid/evaluators.py:183-192 Evaluator(model).evaluate(loader, query, gallery): embed -> query x gallery block -> metrics."""
    import io, contextlib
    import ssg_amd
    from oracle import eval_oracle
    g = golden("embed_ref.npz")
    imgs = torch.randn(4, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    imgs = torch.cat([imgs, imgs.flip(0) * 0.9], 0)           # 8 images, two per "identity"
    names = ["i%d" % i for i in range(8)]; pids = [0, 1, 2, 3, 3, 2, 1, 0]; cams = [0, 0, 0, 0, 1, 1, 1, 1]
    m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=int(g["weight_seed"])).cuda()
    loader = [(imgs, names, pids, cams)]
    query = [(names[i], pids[i], cams[i]) for i in range(4)]; gallery = [(names[i], pids[i], cams[i]) for i in range(8)]
    with contextlib.redirect_stdout(io.StringIO()):
        top1 = ssg_amd.Evaluator(m, print_freq=1).evaluate(loader, query, gallery)
        feats, _ = ssg_amd.extract_features(m, loader)
        dist = ssg_amd.pairwise_distance(feats, query, gallery).numpy()
    _, oscores, otop1 = eval_oracle.evaluate_all(dist, pids[:4], pids, cams[:4], cams)
    assert top1 == otop1


# ------------------------------------------------------------------ original distance: int8 exact Gram / fp64 fallback
@pytest.mark.parametrize("case", [
    # name, N, d, feature scale, env
    ("i8_3digits", 700, 128, 1.0, {}),                                   # unit-norm rows: |feat| <= 0.49 -> 3 digits, 2 k blocks per stage
    ("i8_3digits_odd_blocks", 300, 72, 1.0, {}),                         # d = 72 -> 3 k blocks (one per stage), zero-padded last block
    ("i8_4digits_forced", 500, 64, 1.0, {"SSG_SELF_GRAM_DIGITS": "4"}),
    ("i8_4digits_auto", 400, 40, 3.0, {}),                               # max |feat| in (0.49, 1]
    ("fp64_fallback_range", 400, 64, 9.0, {}),                           # |feat| > 1: the integer path does not apply
    ("fp64_forced", 500, 96, 1.0, {"SSG_SELF_GRAM": "f64"}),
])
def test_self_distance_paths_vs_oracle(case, dev, ora, monkeypatch):
    """Every way the half 'original distance' (rerank.py:33,61-62) can be produced -- exact integer Gram on the int8 matrix
    cores with 3 or 4 digits, fp64-MFMA fallback -- against the oracle's cdist restatement, bit for bit, incl. duplicate rows."""
    from ssg_amd import rerank
    name, N, d, scale, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(len(name))
    x = rng.standard_normal((N, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    x *= np.float32(scale)
    if scale == 3.0:
        x = np.clip(x, -1.0, 1.0); x[0, 0] = 0.9
    x[7] = x[3]                                                          # duplicate rows: distance exactly 0
    h = rerank.re_ranking_device(torch.from_numpy(x[:50].copy()).to(dev), torch.from_numpy(x).to(dev), no_rerank=True)
    got = h.euclid.cpu().numpy().view(np.uint16)
    ref = ora.euclid(x).view(np.uint16)
    assert np.array_equal(got, ref), (name, int((got != ref).sum()))
    assert got[7, 3] == 0 and got[3, 7] == 0


# ------------------------------------------------------------------ kNN-set Jaccard variant
def test_rerank_plain_vs_reference_golden(golden, dev):
    """HIP rerank_plain.re_ranking vs the reference's float64 final_dist (bitwise), eps and DBSCAN labels on top of it."""
    import io, contextlib
    import ssg_amd
    from ssg_amd import cluster, rerank_plain
    g = golden("rerank_plain.npz")
    for tag in "abc":
        k, lam = int(g["k_" + tag]), float(g["lam_" + tag])
        st = {}
        h = rerank_plain.re_ranking_plain_device(torch.from_numpy(g["src_" + tag]).to(dev), torch.from_numpy(g["tgt_" + tag]).to(dev), k=k, lambda_value=lam, stages=st)
        assert np.array_equal(st["a_nnz"].cpu().numpy(), g["setsize_" + tag])
        assert np.array_equal(h.final_dist().cpu().numpy(), g["final_" + tag]), tag
        eps, _, _ = cluster.eps_rule(h, float(g["rho_" + tag]))
        assert eps == float(g["eps_" + tag])
        assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), g["labels_" + tag])
    with contextlib.redirect_stdout(io.StringIO()):
        f1, f2 = ssg_amd.re_ranking_plain(g["src_a"], g["tgt_a"], k=int(g["k_a"]), lambda_value=float(g["lam_a"]))
    assert f1 is f2 and np.array_equal(np.asarray(f1), g["final_a"])


def test_rerank_plain_ties_and_chunks_vs_oracle(dev, ora):
    """Heavy ties at the k-th distance (quantised features -> sets far larger than k, capacity retry), duplicate rows, k=1
    (possibly empty sets) and N > 32768 columns per LDS pass are not needed for correctness of one chunk but the multi-row
    persistent path is: N = 3000."""
    from ssg_amd import rerank_plain
    rng = np.random.default_rng(9)
    N, d = 3000, 32
    tgt = (rng.integers(0, 3, (N, d)) / 8.0).astype(np.float32)          # few distinct distances: large tie groups
    tgt[11] = tgt[4]
    src = rng.standard_normal((200, d)).astype(np.float32) * 0.3
    for k, lam in ((20, 0.1), (1, 0.3)):
        h = rerank_plain.re_ranking_plain_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k=k, lambda_value=lam)
        ref, _ = ora.re_ranking_plain(src, tgt, k=k, lambda_value=lam)
        assert np.array_equal(h.final_dist().cpu().numpy(), ref), (k, lam)


# ------------------------------------------------------------------ split-half embedding on checkpoint-like weights
def _checkpoint_like_state_dict(seed):
    """tools/synth.checkpoint_like_state_dict: Kaiming convolutions + BatchNorm statistics with the spread of a trained, folded checkpoint"""
    from synth import checkpoint_like_state_dict
    return checkpoint_like_state_dict(seed)


def test_split_half_on_checkpoint_like_weights(dev):
    """Per-output-channel weight scales: the split-half embedding stays fp32-class when the folded per-channel BN scales of
    a layer span more than 10^3 (one per-layer power of two would push the small rows into the half subnormals)."""
    import warnings
    import ssg_amd
    from oracle import embed_oracle
    sd = _checkpoint_like_state_dict(7)
    scales = (sd["base.layer2.0.bn2.weight"] / torch.sqrt(sd["base.layer2.0.bn2.running_var"] + 1e-5)).abs()
    assert float(scales.max() / scales.min()) > 1e3
    imgs = torch.randn(3, 3, 256, 128, generator=torch.Generator().manual_seed(11))
    ref = torch.stack(embed_oracle.embed_with_flip(sd, imgs, 2))     # torch fp32 on the CPU, [3, B, 2048]
    errs = {}
    for precision in ("split", "f32"):
        m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, pretrained=False, precision=precision).cuda().eval()
        m.load_state_dict(sd, strict=False)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                      # no overflow fallback may be needed here
            got = m.embed_with_flip(imgs).cpu()
        assert got.shape == ref.shape and bool(torch.isfinite(got).all())
        errs[precision] = float((got - ref).abs().max())
    print("checkpoint-like weights: max |err| vs torch fp32 CPU: split %.3g, f32 %.3g" % (errs["split"], errs["f32"]))
    assert errs["f32"] < 1e-5 and errs["split"] < 1e-5 and errs["split"] < 3.0 * errs["f32"] + 2e-7


@pytest.mark.parametrize("B,H", [(3, 256), (2, 96), (2, 104)])
def test_fused_stem_and_bottleneck_match_the_separate_launches(B, H, dev, monkeypatch):
    """ssg_stem_pool_nchw_x (image -> conv1 + bn + relu -> maxpool in one launch) and ssg_bottleneck[_ds]_nhwc_x (a whole layer1
    block in one launch) against the launch-per-layer path they replace: same k-steps, product order and epilogues, so the
    layer4 map must be bit-identical.  H = 104: ragged last strip of the stem, layer1 / layer2 heights 26 / 13 have no fused block
    kernel (fall back per block); H = 96: short images, layer1 height 24 is fused, layer2 height 12 falls back."""
    import ssg_amd
    from ssg_amd import _lib
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, seed=5).cuda().eval()
    imgs = torch.randn(B, 3, H, 128, generator=torch.Generator().manual_seed(21)).cuda()
    L = _lib.lib()
    assert L.ssg_stem_pool_supported(H, 128) == 1 and L.ssg_bottleneck_supported(H // 4, 32, 256, 256, 64) == (1 if (H // 4) % 4 == 0 else 0)
    assert L.ssg_bottleneck_supported(H // 4, 32, 64, 256, 64) == L.ssg_bottleneck_supported(H // 4, 32, 256, 256, 64)
    assert L.ssg_bottleneck_supported(32, 16, 512, 512, 128) == 1 and L.ssg_bottleneck_supported(26, 16, 512, 512, 128) == 0   # layer2 identity blocks: 8-row tiles
    assert L.ssg_bottleneck_supported(16, 8, 1024, 1024, 256) == 0        # layer3 / layer4: separate launches
    maps = {}
    for stem, bneck in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SSG_FUSED_STEM", stem); monkeypatch.setenv("SSG_FUSED_BOTTLENECK", bneck)
        for flip in (False, True):
            y, sp = m._fmap(imgs, flip=flip)
            assert sp and not m._overflowed()
            maps[(stem, bneck, flip)] = y.clone()
    for flip in (False, True):
        ref = maps[("0", "0", flip)]
        for k in (("1", "1"), ("0", "1"), ("1", "0")):
            assert torch.equal(maps[k + (flip,)].view(torch.int32), ref.view(torch.int32)), (k, flip)
    assert not torch.equal(maps[("1", "1", False)], maps[("1", "1", True)])
    # the fused kernels raise the same range flag as the separate launches
    big = 3.0e4 * imgs
    m._fmap(big); assert m._overflowed()
    with pytest.raises(ValueError, match="unsupported block"):       # shapes without a fused kernel are refused, not mis-run
        x = torch.zeros(1, 16, 8, 1024, device="cuda")
        blk = m._prepare()["blocks"][8]
        _lib.check(L.ssg_bottleneck_nhwc_x(_lib.ptr(x), _lib.ptr(blk["c1"].w), _lib.ptr(blk["c1"].bias), _lib.ptr(blk["c1"].cscale), _lib.ptr(blk["c2"].w),
                                           _lib.ptr(blk["c2"].bias), _lib.ptr(blk["c2"].cscale), _lib.ptr(blk["c3"].w), _lib.ptr(blk["c3"].bias),
                                           _lib.ptr(blk["c3"].cscale), _lib.ptr(torch.empty_like(x)), 1, 16, 8, 1024, 256, None, _lib.stream()), "bottleneck")


def test_split_half_overflow_falls_back_to_f32(dev):
    """Activations beyond the half range (|v| >= 65520): every entry point notices (device flag raised by the encoding
    epilogue), warns once and returns the fp32-path result instead of inf / NaN / clipped features."""
    import ssg_amd
    sd = ssg_amd.synthetic_state_dict(seed=1)
    imgs = 3.0e4 * torch.randn(2, 3, 256, 128, generator=torch.Generator().manual_seed(3))      # stem outputs ~ 1e5
    ms = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False, seed=1, precision="split").cuda().eval()
    m32 = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False, seed=1, precision="f32").cuda().eval()
    with pytest.warns(UserWarning, match="half range"):
        got = ms.embed_with_flip(imgs)
    want = m32.embed_with_flip(imgs)
    assert bool(torch.isfinite(got).all()) and torch.equal(got, want)
    x1, _ = ms(imgs, False); y1, _ = m32(imgs, False)
    assert torch.equal(x1, y1)
    assert torch.equal(ms.feature_map(imgs), m32.feature_map(imgs))
    # extraction reads the flag once after its last batch and recomputes the whole extraction on the fp32 path
    from ssg_amd import evaluators as ev
    ms2 = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False, seed=1, precision="split").cuda().eval()
    both = torch.cat([imgs * 1e-4, imgs])                     # first batch in range, second one overflows
    with pytest.warns(UserWarning, match="half range"):
        fe, _, _ = ev.extract_embeddings(ms2, ev.TensorBatchLoader(both, 2))
    fw, _, _ = ev.extract_embeddings(m32, ev.TensorBatchLoader(both, 2))
    assert bool(torch.isfinite(fe).all()) and torch.equal(fe, fw)
    # in-range batches keep using the split path afterwards (flag cleared)
    small = torch.randn(2, 3, 256, 128, generator=torch.Generator().manual_seed(4))
    a = ms.embed_with_flip(small); b = m32.embed_with_flip(small)
    assert not torch.equal(a, b) and float((a - b).abs().max()) < 5e-6
    del sd


# ------------------------------------------------------------------ epsilon rule: sampled fast path vs radix select vs numpy
def _eps_numpy(M, rho):
    tri = np.triu(M, 1)                      # selftraining.py:289-293
    tri = tri[np.nonzero(tri)]
    tri = np.sort(tri, axis=None)
    top = int(np.round(rho * tri.size))
    return tri[:top].mean(), tri.size, top


@pytest.mark.parametrize("case", ["uniform", "zeros", "unrepresentative_sample", "hidden_small_rows", "half_matrix"])
def test_eps_rule_paths_agree_with_numpy(case, dev, monkeypatch):
    """The sampled-threshold fast path is exact whenever it accepts its result and falls back to the radix select when the
    strided row sample misjudges the quantile: both paths == numpy on matrices built to break the sample."""
    from ssg_amd import cluster
    rng = np.random.default_rng(9)
    N, rho = 1500, 1.6e-3
    M = rng.random((N, N)) + 0.05
    if case == "zeros":
        M[rng.random((N, N)) < 0.3] = 0.0
    elif case == "unrepresentative_sample":        # sampled rows (0, stride, ...) look small, the rest is large: threshold too low
        stride = max(1, N // 192)
        M[::stride] *= 1e-3
    elif case == "hidden_small_rows":               # the small values sit only in rows the sample never sees
        stride = max(1, N // 192)
        M[1::stride] *= 1e-3 if stride > 1 else 1.0
        M[N // 2:] += 5.0
    M = np.triu(M, 1); M = M + M.T
    if case == "half_matrix":
        Mh = M.astype(np.float16)
        ref = _eps_numpy(Mh, rho)
        for path in ("sampled", "radix"):
            monkeypatch.setenv("SSG_EPS_PATH", path)
            eps, cnt, top = cluster.eps_rule(Mh, rho)
            assert (np.float16(eps).view(np.uint16), cnt, top) == (np.float16(ref[0]).view(np.uint16), ref[1], ref[2]), path
        return
    ref = _eps_numpy(M, rho)
    for path in ("sampled", "radix"):
        monkeypatch.setenv("SSG_EPS_PATH", path)
        eps, cnt, top = cluster.eps_rule(M, rho)
        assert (eps, cnt, top) == (float(ref[0]), ref[1], ref[2]), (case, path)


def test_conv_dma_kernel_is_deterministic(L, dev):
    """Race screen for the LDS-DMA pipeline of conv_dma_kernel (counted vmcnt + raw s_barrier): the same launch repeated must give
    the same bits, on the shape and tile (conv3 + residual, 128 x 256 tiles, two workgroups per CU) that exposed a missing
    lgkmcnt(0) before the barrier in round 3 (two wrong output tiles in 1600 per launch)."""
    from ssg_amd._lib import check, ptr, stream
    from ssg_amd.resnet import _h8l8, _weight_scale, pack_weight_khwc
    B, H, W, Cin, Cout = 400, 16, 8, 256, 1024
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
    wk = pack_weight_khwc(w.permute(0, 2, 3, 1)); sc = _weight_scale(wk); ws = _h8l8(wk * sc).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
    zs = torch.zeros(B, H, W, Cout, device=dev)

    def run(res):
        out = torch.empty(B, H, W, Cout, device=dev)
        check(L.ssg_conv2d_nhwc_x(ptr(xs), ptr(ws), ptr(bias), ptr(res), ptr(out), B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 3, 1.0 / sc, None, None, stream()), "convx")
        return out
    ref = run(None)
    for rep in range(12):
        assert torch.equal(run(zs).view(torch.int32), ref.view(torch.int32)), "launch %d with a zero residual differs from the launch without one" % rep
        assert torch.equal(run(None).view(torch.int32), ref.view(torch.int32)), rep


# ------------------------------------------------------------------ JPEG decode on the GPU
def test_jpeg_decode_vs_pillow_golden(golden, dev):
    """csrc/jpeg.hip through ssg_amd.jpeg.decode_batch: the committed files (written and decoded by Pillow, the reference's codec:
    preprocessor.py:28) come back with exactly Pillow's bytes -- 4:4:4 / 4:2:2 / 4:2:0, grayscale, restart intervals, optimised
    Huffman tables, odd sizes -- decoded as ONE batch; the progressive file of the fixture takes the Pillow fallback."""
    from ssg_amd import jpeg as pj
    g = golden("jpeg_cases.npz")
    n = int(g["count"])
    files = [g["file_%02d" % i].tobytes() for i in range(n)] + [g["progressive_file"].tobytes()]
    before = dict(pj.stats)
    out = pj.decode_batch(files)
    assert pj.stats["gpu"] - before["gpu"] == n and pj.stats["pillow"] - before["pillow"] == 1
    for i in range(n):
        assert out[i].dtype == torch.uint8 and out[i].is_cuda
        assert np.array_equal(out[i].cpu().numpy(), g["rgb_%02d" % i]), i
    assert np.array_equal(out[n].cpu().numpy(), g["progressive_rgb"])


def test_jpeg_damaged_files_behave_like_pillow(golden, dev):
    """damaged entropy-coded data must not come back as plausible pixels.  A file cut short inside its scan is flagged by the
    Huffman kernel's per-image status word and handed to Pillow, which raises "image file is truncated" exactly like the reference's
    Image.open(...).convert('RGB'); a file whose scan is short but still ends in EOI decodes to what Pillow produces (libjpeg feeds
    zeros); the undamaged files of the same batch are unaffected."""
    import io
    from PIL import Image
    from ssg_amd import jpeg as pj
    g = golden("jpeg_cases.npz")
    good = [g["file_%02d" % i].tobytes() for i in range(4)]
    victim = good[1]
    h = pj.scan_header(victim)
    cut = victim[:h.ecs_start + (h.ecs_end - h.ecs_start) // 2]
    with pytest.raises(OSError):
        Image.open(io.BytesIO(cut)).convert("RGB")
    with pytest.raises(OSError):
        pj.decode_batch([good[0], cut, good[2]])
    short = cut + b"\xff\xd9"                   # half the scan, then EOI: libjpeg pads with zeros and warns, Pillow returns pixels
    try:
        ref = np.asarray(Image.open(io.BytesIO(short)).convert("RGB"))
    except OSError:
        ref = None
    before = pj.stats.get("damaged", 0)
    if ref is None:
        with pytest.raises(OSError):
            pj.decode_batch([good[0], short, good[2]])
    else:
        out = pj.decode_batch([good[0], short, good[2]])
        assert np.array_equal(out[1].cpu().numpy(), ref)
        assert np.array_equal(out[0].cpu().numpy(), g["rgb_00"]) and np.array_equal(out[2].cpu().numpy(), g["rgb_02"])
    assert pj.stats.get("damaged", 0) == before + 1


def test_jpeg_decode_generated_files_and_loader(dev, ora, tmp_path):
    """Freshly generated files (Pillow as the checker at run time) in one batch of mixed sizes and layouts, and the extraction
    loader end to end: GpuBatchLoader(decode='gpu') == GpuBatchLoader(decode='pillow') bit for bit on files on disk."""
    import io
    from PIL import Image
    import ssg_amd
    from ssg_amd import jpeg as pj
    rng = np.random.default_rng(9)
    files, refs = [], []
    for k in range(40):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([xx * 255.0 / max(w - 1, 1), yy * 255.0 / max(h - 1, 1), (xx + yy) * 127.0 / max(h + w - 2, 1)], -1) + rng.normal(0, 25, (h, w, 3))
        kw = dict(quality=int(rng.integers(25, 99)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
        if rng.integers(0, 3) == 0:
            kw["restart_marker_blocks"] = int(rng.integers(1, 6))
        buf = io.BytesIO(); Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(buf, "JPEG", **kw)
        files.append(buf.getvalue()); refs.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    out = pj.decode_batch(files)
    for k in range(40):
        assert np.array_equal(out[k].cpu().numpy(), refs[k]), (k, refs[k].shape)
    # the loader: Market-1501-like 128 x 64 files plus a few other sizes and one PNG (Pillow fallback inside the GPU path)
    names = []
    for k in range(11):
        h, w = (128, 64) if k < 8 else (int(rng.integers(60, 200)), int(rng.integers(30, 100)))
        a = np.clip(rng.normal(120, 50, (h, w, 3)), 0, 255).astype(np.uint8)
        name = "img%02d.%s" % (k, "png" if k == 10 else "jpg")
        Image.fromarray(a).save(str(tmp_path / name), quality=85) if k != 10 else Image.fromarray(a).save(str(tmp_path / name))
        names.append((name, k % 3, 0))
    got = list(ssg_amd.GpuBatchLoader(names, root=str(tmp_path), height=256, width=128, batch_size=4, decode="gpu"))
    ref = list(ssg_amd.GpuBatchLoader(names, root=str(tmp_path), height=256, width=128, batch_size=4, decode="pillow"))
    assert len(got) == len(ref) == 3
    for (a, fa, pa, _), (b, fb, pb, _) in zip(got, ref):
        assert fa == fb and pa == pb and torch.equal(a, b)
