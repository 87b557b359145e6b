"""GPU suite (-m gpu): the embed -> grouping hand-off, end to end (VERDICT r2 missing #3).

The stage tests feed synthetic embeddings to the grouping kernels and synthetic images to the embedder separately; here the
float32 features that come OUT of the HIP ResNet-50 go INTO the HIP distance / re-rank / eps / DBSCAN path, and the oracle
(CPU restatement of reid/rerank.py + selftraining.py:289-306 + sklearn DBSCAN) runs on those same features: eps and labels must
be identical, distances bit for bit.

* BASELINE configs[0]: N = 2 000 synthetic 256x128 Track-I images (N(0,1) pixels, seed 1) ->
  extract_features -> no-rerank squared L2 (half) -> eps rule -> DBSCAN, through the selftraining.py call surface
  (`compute_dist` -> `generate_selflabel`, selftraining.py:196-219,255-313);
* configs[1] on embedder output: the same chain at N = 16 000 needs 16 000 images through the embedder (~0.7 s) and the oracle's
  no-rerank path (seconds);
* configs[2] on embedder output with checkpoint-like BatchNorm statistics and images that carry identities (so that the
  features are not degenerate and the k-reciprocal sets, the Jaccard rows and the clusters are non-trivial).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda", 0)


def identity_images(n, n_id, seed, noise=0.35, device="cpu"):
    """images that carry identities: a random 3x16x8 pattern per identity, upsampled x16 (nearest), plus per-image pixel noise"""
    g = torch.Generator(device=device).manual_seed(seed)
    pat = torch.randn(n_id, 3, 16, 8, generator=g, device=device)
    ids = torch.arange(n, device=device) % n_id
    base = pat[ids].repeat_interleave(16, dim=2).repeat_interleave(16, dim=3)
    return base + noise * torch.randn(n, 3, 256, 128, generator=g, device=device), ids.cpu().numpy()


def _features(model, imgs, batch=250):
    import ssg_amd
    feats, names, _ = ssg_amd.extract_embeddings(model, ssg_amd.TensorBatchLoader(imgs, batch), for_eval=False)
    return feats, names


def test_config0_track_i_chain_vs_oracle(dev, ora, capsys):
    """BASELINE configs[0] (N = 2 000, plumbing case): Track-I images -> HIP embed -> no-rerank L2 -> eps -> DBSCAN == oracle on the
    same HIP features, through extract_features / compute_dist / generate_selflabel exactly as selftraining.py:196-219 calls them."""
    import ssg_amd
    ora.set_num_threads(min(os.cpu_count() or 8, 64))
    N, Ns = 2000, 500
    g = torch.Generator(device=dev).manual_seed(1)
    tgt_imgs = torch.randn(N, 3, 256, 128, generator=g, device=dev)
    src_imgs = torch.randn(Ns, 3, 256, 128, generator=g, device=dev)
    model = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False).cuda().eval()
    names = ["t%05d.jpg" % i for i in range(N)]
    # the dictionary surface of the reference (evaluators.py:18-60), then its reorder + stack (selftraining.py:197-209)
    tf, _ = ssg_amd.extract_features(model, ssg_amd.TensorBatchLoader(tgt_imgs, 128, names), print_freq=1000, for_eval=False)
    target_features = torch.cat([tf[f].unsqueeze(0) for f in names], 0)
    sf, _ = ssg_amd.extract_features(model, ssg_amd.TensorBatchLoader(src_imgs, 128), print_freq=1000, for_eval=False)
    source_features = torch.cat([v.unsqueeze(0) for v in sf.values()], 0)
    assert target_features.shape == (N, 2048) and bool(torch.isfinite(target_features).all())
    args = SimpleNamespace(no_rerank=True, rho=1.6e-3)
    e_list, r_list = ssg_amd.compute_dist(source_features, target_features, lambda_value=0.1, no_rerank=True, num_split=1)
    labels, clusters = ssg_amd.generate_selflabel(e_list, r_list, 0, args, [])
    tgt_np = target_features.numpy()
    oe, _ = ora.re_ranking(source_features.numpy(), tgt_np, no_rerank=True)
    assert np.array_equal(bits(e_list[0].euclid.cpu().numpy()), bits(oe)), "euclidean_dist on embedder output"
    oeps, ocnt, otop = ora.eps_rule(oe, 1.6e-3)
    assert bits(np.float16(clusters[0].eps)) == bits(np.float16(oeps)), "eps"
    assert np.array_equal(labels[0], ora.dbscan(oe, oeps, 4)), "labels"
    # iteration 1 reuses the cached estimator (eps frozen, selftraining.py:297-298)
    labels1, clusters1 = ssg_amd.generate_selflabel(e_list, r_list, 1, args, clusters)
    assert clusters1[0] is clusters[0] and np.array_equal(labels1[0], labels[0])
    d2 = oe.astype(np.float64)
    print("configs[0]: N=%d Track-I features: median d^2 %.3g (degenerate, SURVEY 7.5), eps %.4g, %d clusters, %d noise" % (
        N, float(np.median(d2)), float(oeps), labels[0].max() + 1, int((labels[0] < 0).sum())))


@pytest.fixture(scope="module")
def identity_features(dev):
    """HIP embeddings of identity-carrying images under checkpoint-like BatchNorm statistics: N = 16 000 targets + 4 000 sources"""
    import ssg_amd
    from test_gpu_parity import _checkpoint_like_state_dict
    model = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False).cuda().eval()
    model.load_state_dict(_checkpoint_like_state_dict(7), strict=False)
    N, Ns = 16000, 4000
    tgt = []
    for c0 in range(0, N, 4000):                       # 4000 images = 1.6 GB at a time
        imgs, _ = identity_images(4000, 250, 100 + c0, device=dev)
        # identities differ from chunk to chunk: 1000 identities of 16 images
        tgt.append(_features(model, imgs)[0])
    simgs, _ = identity_images(Ns, 400, 999, noise=0.5, device=dev)
    src = _features(model, simgs)[0]
    tgt = torch.cat(tgt, 0)
    assert tgt.shape == (N, 2048) and bool(torch.isfinite(tgt).all())
    return src, tgt


def test_config1_norerank_on_embedder_output_vs_oracle(identity_features, ora):
    """BASELINE configs[1] at the headline size on real embedder output: N = 16 000 HIP features -> pairwise L2 (half) -> eps ->
    DBSCAN, eps and labels == oracle on the same features (at N = 16 000 only euclidean_dist bits were checked)."""
    from ssg_amd import rerank, cluster
    ora.set_num_threads(min(os.cpu_count() or 8, 64))
    src, tgt = identity_features
    h = rerank.re_ranking_device(src, tgt, no_rerank=True)
    oe, _ = ora.re_ranking(src.cpu().numpy(), tgt.cpu().numpy(), no_rerank=True)
    assert np.array_equal(bits(h.euclid.cpu().numpy()), bits(oe)), "euclidean_dist"
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    oeps, ocnt, otop = ora.eps_rule(oe, 1.6e-3)
    assert (bits(np.float16(eps)), cnt, top) == (bits(np.float16(oeps)), ocnt, otop), "eps rule (half matrix)"
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(h)
    assert np.array_equal(lab, ora.dbscan(oe, oeps, 4)), "labels"
    print("configs[1] on embedder output: eps %.4g, %d clusters, %d noise" % (float(eps), lab.max() + 1, int((lab < 0).sum())))


def test_config2_rerank_on_embedder_output_vs_oracle(identity_features, ora):
    """BASELINE configs[2] on real embedder output (checkpoint-like weights, identity-carrying images): the float32 features ->
    half rounding -> int8 digits hand-off, source term, k-reciprocal re-rank, eps, DBSCAN -- every array == oracle."""
    from ssg_amd import rerank, cluster
    ora.set_num_threads(min(os.cpu_count() or 8, 64))
    src, tgt = identity_features
    N = 6000                                            # the oracle's float64 cdist at d = 2048: a few seconds at this size
    tgt = tgt[:N].contiguous()
    st = {}
    h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, stages=st)
    oe, of, ost = ora.re_ranking(src.cpu().numpy(), tgt.cpu().numpy(), k1=20, k2=6, lambda_value=0.3, stages=True)
    assert np.array_equal(bits(st["D"].cpu().numpy()), bits(oe)), "original distance"
    assert np.array_equal(bits(st["v"].cpu().numpy()), bits(ost["v"])), "source vector"
    assert np.array_equal(st["rank"].cpu().numpy(), ost["rank"]), "initial rank (introsort tie order)"
    assert np.array_equal(bits(st["Jp"].cpu().numpy()), bits(ost["jaccard_scaled"])), "scaled jaccard"
    assert np.array_equal(h.final_dist().cpu().numpy(), of), "final_dist"
    eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
    assert (eps, cnt, top) == ora.eps_rule(of, 1.6e-3), "eps rule"
    lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
    assert np.array_equal(lab, ora.dbscan(of, eps, 4)), "labels"
    # the features are not degenerate: most identities come back as clusters
    ncl = int(lab.max() + 1)
    print("configs[2] on embedder output: N=%d eps %.4f, %d clusters, %d noise" % (N, eps, ncl, int((lab < 0).sum())))
    assert ncl > 50


def test_f32_twin_follows_load_state_dict(dev):
    """the cached fp32 fallback model must not survive load_state_dict() / cuda() -- INTEGRATION.md's loop
    loads new weights every self-training iteration.  Overflow, load new weights, overflow again: the second result must equal a
    fresh precision='f32' model with the NEW weights."""
    import warnings
    import ssg_amd
    imgs = torch.randn(2, 3, 256, 128, generator=torch.Generator().manual_seed(5)) * 3.0e4      # stem outputs leave the half range
    m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False).cuda().eval()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = m.embed_with_flip(imgs)
        assert m._twin is not None, "the test input must trigger the fp32 fallback"
        sd2 = ssg_amd.synthetic_state_dict(seed=2)
        m.load_state_dict(sd2, strict=False)
        assert m._twin is None
        b = m.embed_with_flip(imgs)
        fresh = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False, precision="f32").cuda().eval()
        fresh.load_state_dict(sd2, strict=False)
        ref = fresh.embed_with_flip(imgs)
    assert torch.equal(b, ref), "fallback after load_state_dict used stale weights"
    assert not torch.equal(a, b)
