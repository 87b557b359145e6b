"""GPU suite (-m gpu), BASELINE.json sizes: the feature dimension (d = 2048) and the set sizes the bench runs.

* d = 2048 against the oracle at every stage boundary (the int8 Gram's digit paths, the LDS-DMA source-term
  kernel with padded / non-256-multiple source counts);
* configs[1]+[2]: N = 16 000, Ns = 12 936, d = 2048 -- the whole grouping step (distances, eps, labels) bit for bit
  against the oracle, on the survey's Track-G set and on the hard set (noise, border points, large edge lists);
* configs[3] (N = 30 000 x 3 splits) and configs[4] (N = 128 000): size-independent properties plus numpy itself as the
  checker on sampled rows (np.argsort of the normalised half row == the device ranking; float64 restatement of
  final_dist on sampled entries).
"""
import os

import numpy as np
import pytest

from conftest import bits, clustered, hard_clustered

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda", 0)


def _oracle_threads(ora):
    n = min(os.cpu_count() or 8, 64)
    ora.set_num_threads(n)
    return n


def _sparse_rows_equal(idx, val, nnz, dense_bits):
    """sparse rows (sorted columns) == the non-zeros of a dense half matrix, without densifying on the host per row"""
    idx = idx.cpu().numpy(); val = val.cpu().numpy().view(np.uint16); nnz = nnz.cpu().numpy()
    N = dense_bits.shape[1]
    out = np.zeros_like(dense_bits)
    rows = np.repeat(np.arange(idx.shape[0]), nnz)
    cols = np.concatenate([idx[i, :nnz[i]] for i in range(idx.shape[0])]) if len(rows) else np.zeros(0, np.int64)
    vals = np.concatenate([val[i, :nnz[i]] for i in range(idx.shape[0])]) if len(rows) else np.zeros(0, np.uint16)
    assert cols.min() >= 0 and cols.max() < N
    out[rows, cols] = vals
    return np.array_equal(out, dense_bits)


@pytest.mark.parametrize("N,Ns", [(1500, 1000), (1333, 777), (600, 2049)])
def test_bench_dimension_stages_vs_oracle(N, Ns, dev, ora):
    """d = 2048 (the embedding width of the bench): D, v, rank, V, V_qe, J', final, eps, labels == oracle."""
    from ssg_amd import rerank, cluster
    _oracle_threads(ora)
    d = 2048
    tgt = hard_clustered(N, d, 11); src = hard_clustered(Ns, d, 12, intra=0.7)
    tgt[5] = tgt[3]
    st = {}
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=0.3, stages=st)
    oe, of, ost = ora.re_ranking(src, tgt, lambda_value=0.3, stages=True)
    assert np.array_equal(bits(st["D"].cpu().numpy()), bits(oe)), "original distance (int8 Gram, d=2048)"
    assert np.array_equal(bits(st["v"].cpu().numpy()), bits(ost["v"])), "source vector (LDS-DMA bound pass + float64 refine)"
    assert np.array_equal(st["rank"].cpu().numpy(), ost["rank"]), "initial rank"
    assert _sparse_rows_equal(st["v_idx"], st["v_val"], st["v_nnz"], bits(ost["V"])), "V"
    assert _sparse_rows_equal(st["q_idx"], st["q_val"], st["q_nnz"], bits(ost["V_qe"])), "V_qe"
    assert np.array_equal(bits(st["Jp"].cpu().numpy()), bits(ost["jaccard_scaled"])), "scaled jaccard"
    assert np.array_equal(h.final_dist().cpu().numpy(), of), "final_dist"
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    assert (eps, cnt, top) == ora.eps_rule(of, 1.6e-3)
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h), ora.dbscan(of, eps, 4))
    # the fp64 fallback of the self distance and the exhaustive source term give the same bits at this width
    os.environ["SSG_SELF_GRAM"] = "f64"
    try:
        h2 = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), no_rerank=True)
    finally:
        del os.environ["SSG_SELF_GRAM"]
    assert torch.equal(h2.euclid, st["D"])
    t = rerank._as_dev_f32(tgt, dev); s = rerank._as_dev_f32(src, dev)
    assert torch.equal(rerank.source_vector(s, t), rerank.source_vector(s, t, exact_gemm=True))


def test_bench_width_vs_reference_golden(golden, dev):
    """the HIP path against the UNTOUCHED reference at the bench's feature width -- tests/golden/rerank_wide_d2048_ref.npz
    (tools/make_golden.py --only-wide: reid/rerank.py re_ranking at N = Ns = 2000, d = 2048, hard set, lambda = 0.3, numpy's default
    argsort; eps rule of selftraining.py:289-293; sklearn DBSCAN).  No oracle in between: euclidean_dist and final_dist by sha256 of
    the materialised matrices, the rank columns, the source vector, eps, labels."""
    import hashlib
    from ssg_amd import rerank, cluster
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    g = golden("rerank_wide_d2048_ref.npz")
    N, Ns, d = int(g["N"]), int(g["Ns"]), int(g["d"])
    tgt = hard_clustered(N, d, int(g["seed_tgt"])); src = hard_clustered(Ns, d, int(g["seed_src"]), intra=float(g["intra_src"]))
    assert sha(tgt) == str(g["sha_tgt"]) and sha(src) == str(g["sha_src"])
    st = {}
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k1=int(g["k1"]), k2=int(g["k2"]),
                                 lambda_value=float(g["lambda_value"]), stages=st)
    assert sha(h.euclid.cpu().numpy()) == str(g["sha_euclid"]), "euclidean_dist (int8 Gram at d = 2048) vs the reference's cdist"
    assert np.array_equal(st["rank"].cpu().numpy()[:, :21], g["rank"]), "initial_rank[:, :21] in numpy's introsort order"
    v = st["v"].cpu().numpy()
    assert np.array_equal((v + v[0]).astype(np.float64), g["v"]), "source vector"
    final = h.final_dist().cpu().numpy()
    assert np.array_equal(final[0], g["final_row0"]) and np.array_equal(np.diag(final), g["final_diag"])
    assert sha(final) == str(g["sha_final"]), "final_dist vs the reference"
    eps, cnt, top = cluster.eps_rule(h, float(g["rho"]))
    assert (eps, cnt, top) == (float(g["eps"]), int(g["count"]), int(g["top_num"]))
    assert np.array_equal(cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(h), g["labels"])
    # the drop-in numpy API on the same inputs
    e2, f2 = rerank.re_ranking(src, tgt, k1=int(g["k1"]), k2=int(g["k2"]), lambda_value=float(g["lambda_value"]))
    assert sha(e2) == str(g["sha_euclid"]) and sha(np.asarray(f2)) == str(g["sha_final"])


@pytest.mark.parametrize("kind", ["track_g", "hard"])
def test_headline_size_labels_vs_oracle(kind, dev, ora):
    """BASELINE configs[1]+[2] at the size the bench is quoted on: N = 16 000 (DukeMTMC-size), Ns = 12 936 (Market
    trainval), d = 2048, k1 = 20, k2 = 6, lambda = 0.3 -- euclidean_dist, final_dist, eps and the DBSCAN labels of the
    device path are bit-identical to the oracle's (the oracle needs ~10-60 s of host cores here)."""
    from ssg_amd import rerank, cluster
    nthr = _oracle_threads(ora)
    N, Ns, d = 16000, 12936, 2048
    gen = clustered if kind == "track_g" else hard_clustered
    tgt = gen(N, d, 1); src = gen(Ns, d, 2, intra=0.7)
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k1=20, k2=6, lambda_value=0.3)
    import time
    t0 = time.time()
    oe, of = ora.re_ranking(src, tgt, k1=20, k2=6, lambda_value=0.3)
    t_ora = time.time() - t0
    assert np.array_equal(bits(h.euclid.cpu().numpy()), bits(oe)), "euclidean_dist"
    final = h.final_dist().cpu().numpy()
    assert np.array_equal(final, of), "final_dist"
    del final
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    oeps, ocnt, otop = ora.eps_rule(of, 1.6e-3)
    assert (eps, cnt, top) == (oeps, ocnt, otop), "eps rule"
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(h)
    olab = ora.dbscan(of, oeps, 4)
    assert np.array_equal(lab, olab), "DBSCAN labels"
    # the device chain the product's generate_selflabel (and the bench) runs -- eps rule -> region query -> components with ONE read-back --
    # against the ORACLE at the headline size (its sample sort works at the upper end of its window here)
    h2 = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k1=20, k2=6, lambda_value=0.3, validate=False)
    e2, c2, t2, l2, core2 = cluster.eps_rule_dbscan(h2, 1.6e-3, min_samples=4)
    assert (e2, c2, t2) == (oeps, ocnt, otop), "fused chain: eps rule vs the oracle"
    assert np.array_equal(l2, olab), "fused chain: labels vs the oracle"
    assert np.array_equal(core2, np.nonzero((of <= oeps).sum(axis=1) >= 4)[0]), "fused chain: core samples vs the oracle"
    del h2
    deg = int((of <= eps).sum())
    print("N=%d %s: oracle re_ranking %.1f s on %d threads; eps %.6f, %d clusters, %d noise, %d region-query hits (%.1f per row)" % (
        N, kind, t_ora, nthr, eps, lab.max() + 1, int((lab < 0).sum()), deg, deg / N))
    if kind == "hard":
        assert (lab < 0).sum() > 0 and lab.max() > 100
        # a radius with ~150 neighbours per row: the edge list outgrows its first allocation (64 N) and the region query retries
        big = float(np.partition(of[::61].ravel(), int(150.0 / N * of[::61].size))[int(150.0 / N * of[::61].size)])
        est = cluster.DBSCAN(eps=big, min_samples=4, metric="precomputed").fit(h)
        nhit = int((of <= big).sum())
        assert nhit > 64 * N
        assert np.array_equal(est.labels_, ora.dbscan(of, big, 4)), "DBSCAN labels at the large radius"
        assert np.array_equal(est.core_sample_indices_, np.nonzero((of <= big).sum(axis=1) >= 4)[0])


def _check_sampled_rows(h, st, rows, k1, lam):
    """numpy as the checker on a few rows of a matrix too large for the oracle: ranking == np.argsort(default kind) of the
    normalised half row (reid/rerank.py:68-70), final_dist == float64 restatement of :122 from J' and v."""
    D = st["D"]; rowmax = st["rowmax"].cpu().numpy().astype(np.uint16).view(np.float16)
    rank = st["rank"].cpu().numpy()
    v = st["v"].cpu().numpy()
    Jp = h.M
    for r in rows:
        drow = D[r].cpu().numpy()
        dn = drow / rowmax[r]                               # half / half -> half, like original_dist / max(axis=0)
        assert np.array_equal(np.argsort(dn)[: k1 + 1].astype(np.int32), rank[r, : k1 + 1]), "row %d ranking vs np.argsort" % r
        jp = Jp[r].cpu().numpy()
        ref = jp.astype(np.float64) + (v + v[r]).astype(np.float64) * lam
        out = torch.empty((1, h.N), dtype=torch.float64, device=Jp.device)
        from ssg_amd._lib import check, lib, ptr, stream
        check(lib().ssg_final_dist_f64(ptr(Jp[r:r + 1]), ptr(h.v), h.N, int(r), 1, lam, ptr(out), stream()), "final")
        assert np.array_equal(out.cpu().numpy()[0], ref), "row %d final_dist" % r


def test_config3_three_splits_properties(dev):
    """BASELINE configs[3]: N = 30 000 (MSMT17-size), 3 feature splits (whole / upper / lower), d = 2048, through
    compute_dist -> generate_selflabel on one GPU (the 8-GPU leg is the same code on row blocks: tests/test_dist.py)."""
    from types import SimpleNamespace
    from ssg_amd import cluster, compute_dist, generate_selflabel, rerank
    N, Ns, d, lam = 30000, 12936, 2048, 0.1
    tgts = [torch.from_numpy(hard_clustered(N, d, 100 + s)).to(dev) for s in range(3)]
    srcs = [torch.from_numpy(hard_clustered(Ns, d, 200 + s, intra=0.7)).to(dev) for s in range(3)]
    e_list, r_list = compute_dist(srcs, tgts, lambda_value=lam, no_rerank=False, num_split=2)
    assert e_list == [[], [], []] and len(r_list) == 3
    args = SimpleNamespace(no_rerank=False, rho=1.6e-3)
    labels, clusters = generate_selflabel(e_list, r_list, 0, args, [])
    for s in range(3):
        h = r_list[s]
        Jp = h.M
        assert Jp.shape == (N, N) and torch.equal(Jp, Jp.T), "J' must be exactly symmetric"
        assert bool((Jp >= 0).all()) and bool((Jp.float() <= 1.0).all())
        lab = labels[s]
        assert lab.shape == (N,) and lab.min() >= -1 and 0 < clusters[s].eps < 1.5
        # the chain's eps (generate_selflabel ran cluster.eps_rule_dbscan: bitonic sort at this size) against the two-call eps rule, and
        # the chain called directly against both
        e_two, c_two, t_two = cluster.eps_rule(h, args.rho)
        assert clusters[s].eps == e_two, "split %d: eps of the chain vs eps_rule" % s
        e_ch, c_ch, t_ch, l_ch, _ = cluster.eps_rule_dbscan(h, args.rho, min_samples=4)
        assert (e_ch, c_ch, t_ch) == (e_two, c_two, t_two) and np.array_equal(l_ch, lab)
        # sklearn numbering: cluster ids in order of their smallest core sample
        est = cluster.DBSCAN(eps=clusters[s].eps, min_samples=4, metric="precomputed").fit(h)
        assert np.array_equal(est.labels_, lab)
        first = {}
        for i in est.core_sample_indices_:
            first.setdefault(int(lab[i]), int(i))
        order = [first[l] for l in sorted(first)]
        assert order == sorted(order) and sorted(first) == list(range(len(first)))
    # one split again with the stage tensors: sampled rows against numpy
    st = {}
    h = rerank.re_ranking_device(srcs[0], tgts[0], lambda_value=lam, stages=st)
    assert torch.equal(h.M, r_list[0].M), "the pipeline is deterministic"
    assert torch.equal(st["D"], st["D"].T) and bool((torch.diagonal(st["D"]) == 0).all())
    _check_sampled_rows(h, st, [0, 1, 7777, 15000, 29999], 20, lam)


def test_config4_n128k_properties(dev):
    """BASELINE configs[4]: N = 128 000 synthetic, d = 2048 on one GPU (the 8-GPU leg shards the same kernels by row block):
    the two N x N half matrices are 32.8 GB each; the ranking runs from the global arena (rows do not fit in LDS)."""
    from ssg_amd import cluster, rerank
    N, Ns, d, lam = 128000, 12936, 2048, 0.1
    tgt = torch.from_numpy(clustered(N, d, 1)).to(dev); src = torch.from_numpy(clustered(Ns, d, 2, intra=0.7)).to(dev)
    st = {}
    h = rerank.re_ranking_device(src, tgt, lambda_value=lam, stages=st)
    rows = sorted(set([0, 63999, 64000, 127999] + [int(r) for r in np.random.default_rng(7).integers(0, N, 64)]))
    _check_sampled_rows(h, st, rows, 20, lam)
    # the streamed replay's rare path: EVERY row it flagged for the in-place kernel (a first pivot inside [0, K)) is
    # checked against np.argsort -- the ranking is re-run with the diagnostic surface to learn which rows those are
    dg = {}
    rank2 = rerank.initial_rank(st["D"], st["rowmax"], N, N, 21, diag=dg)
    assert torch.equal(rank2, st["rank"][:, :21]), "the ranking is deterministic"
    assert dg["flagged"] is not None, "N = 128 000 runs the streamed kernel"
    flagged = torch.nonzero(dg["flagged"]).flatten().cpu().numpy()
    print("N=%d: %d random rows checked; the streamed replay flagged %d rows for the in-place kernel" % (N, len(rows), len(flagged)))
    rowmax_h = st["rowmax"].cpu().numpy().astype(np.uint16).view(np.float16)
    for r in flagged[:512]:
        dn = st["D"][int(r)].cpu().numpy() / rowmax_h[int(r)]
        assert np.array_equal(np.argsort(dn)[:21].astype(np.int32), rank2[int(r)].cpu().numpy()), "flagged row %d vs np.argsort" % r
    del rank2, dg
    Jp = h.M
    # symmetry without a second 32 GB matrix: compare row blocks with the matching column blocks
    for lo in range(0, N, 16000):
        assert torch.equal(Jp[lo:lo + 16000, :].T.contiguous(), Jp[:, lo:lo + 16000].contiguous()), "J' symmetric"
        assert torch.equal(st["D"][lo:lo + 16000, :].T.contiguous(), st["D"][:, lo:lo + 16000].contiguous()), "D symmetric"
    del st
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    assert top == int(np.round(1.6e-3 * cnt)) and 0 < eps < 1.5
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
    # 8000 identities of 16: at this size the rho-quantile radius exceeds the closest centre pairs, so identities may merge
    # (a property of the data, the reference behaves the same); what must hold is that no identity is split or left as noise
    truth = np.arange(N) % (N // 16)
    assert len(np.unique(lab[lab >= 0])) > 1000
    by_id = lab.reshape(16, N // 16)
    assert bool((by_id == by_id[0]).all()) and bool((by_id[0] >= 0).all()), "every identity of the separable Track-G set is kept whole"
