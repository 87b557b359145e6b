"""Small call-surface rows of SURVEY.md 8(a) that the big parity files do not exercise on their own:
a3 extract_cnn_feature, a2 fliplr, a10 the generate_dataloader label join, the DeviceBackedArray handle rules."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")


# ------------------------------------------------------------------ a10: selftraining.py:315-324 (CPU)
def _reference_join(trainval, labels_list):
    """restatement of the reference loop, selftraining.py:315-324 (generate_dataloader): an image is kept iff no split
    labelled it -1; its label is the list of its per-split cluster ids; camera id is replaced by 0"""
    new_dataset = []
    for i, (fname, _, _) in enumerate(trainval):
        label = [labels_list[s][i] for s in range(len(labels_list))]
        if -1 in label:
            continue
        new_dataset.append((fname, label, 0))
    return new_dataset


def test_select_labeled_matches_reference_join():
    from ssg_amd import selftraining
    rng = np.random.default_rng(0)
    for S in (1, 3):
        for n in (0, 1, 57, 400):
            labels_list = [rng.integers(-1, 6, n).astype(np.int64) for _ in range(S)]
            trainval = [("img_%04d.jpg" % i, int(rng.integers(0, 99)), int(rng.integers(0, 6))) for i in range(n)]
            ref = _reference_join(trainval, labels_list)
            keep, lab = selftraining.select_labeled(labels_list)
            got = [(trainval[i][0], [int(x) for x in lab[r]], 0) for r, i in enumerate(keep)]
            assert got == ref
            got2 = selftraining.generate_dataset(trainval, labels_list)
            assert got2 == ref
    keep, lab = selftraining.select_labeled([np.array([-1, -1]), np.array([0, -1])])   # every image dropped
    assert keep.size == 0 and lab.shape == (0, 2)


def test_device_backed_array_never_inherits_its_handle():
    """arithmetic / copies / astype results of a materialised final_dist must not keep the device handle
    (cluster.as_handle would then cluster the ORIGINAL device matrix and ignore the edited values)."""
    from ssg_amd.rerank import DeviceBackedArray
    base = np.arange(16, dtype=np.float64).reshape(4, 4)
    token = object()
    a = DeviceBackedArray.attach(base.copy(), token)
    assert a.valid_handle() is token and not a.flags.writeable
    for b in (-1.0 * a, np.minimum(a, 1), a.copy(), a.astype(np.float32), a[:], a.T, a + 0, np.array(a), a.view(DeviceBackedArray)):
        assert getattr(b, "ssg_handle", None) is None
        if isinstance(b, DeviceBackedArray):
            assert b.valid_handle() is None
    with pytest.raises(ValueError):
        a[a > 3] = 0                       # read-only: in-place edits cannot desynchronise array and device matrix
    a.setflags(write=True)                 # a caller who insists on editing in place gives up the handle
    a[0, 0] = 5.0
    assert a.valid_handle() is None


def test_resnet_pretrained_and_state_dict_diagnostics():
    """pretrained=True without weights and strict=False loads that match nothing must not be silent."""
    import ssg_amd
    with pytest.warns(UserWarning, match="synthetic"):
        m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False)        # reference default pretrained=True
    sd = m.state_dict()
    with pytest.warns(UserWarning, match="backbone tensors"):
        m.load_state_dict({"feat.weight": sd["feat.weight"]}, strict=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m2 = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False, seed=3)
        # nn.DataParallel prefix + the whole checkpoint dict, as selftraining.py:129-132 / serialization.py pass them
        missing, unexpected = m2.load_state_dict({"state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 3}, strict=False)
    assert not missing and not unexpected and m2._weights == "loaded"
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)


# ------------------------------------------------------------------ a2 / a3 on the GPU
@pytest.mark.gpu
def test_extract_cnn_feature_and_fliplr(golden):
    """reid/feature_extraction/cnn.py:10-22 and reid/evaluators.py:12-16: model.eval(), forward, per-split CPU tensors;
    fliplr reverses W.  extract_features' own result (orig + flipped, normalised) must be reproducible from the two."""
    import ssg_amd
    from ssg_amd import evaluators
    g = golden("embed_ref.npz")
    imgs = torch.randn(4, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    fl = evaluators.fliplr(imgs)
    assert fl.shape == imgs.shape and torch.equal(fl, imgs.flip(3)) and torch.equal(evaluators.fliplr(fl), imgs)
    for S in (1, 2):
        m = ssg_amd.create("resnet50", num_classes=0, num_split=S, cluster=False, seed=int(g["weight_seed"]), pretrained=False).cuda()
        out = evaluators.extract_cnn_feature(m, imgs, False)
        out_f = evaluators.extract_cnn_feature(m, fl, False)
        if S == 1:
            assert torch.is_tensor(out) and out.device.type == "cpu" and out.shape == (4, 2048)
            out, out_f = [out], [out_f]
        else:
            assert isinstance(out, list) and len(out) == S + 1 and all(o.device.type == "cpu" and o.shape == (4, 2048) for o in out)
        ref = g["feats_S%d" % S]
        for s in range(len(out)):
            x = out[s] + out_f[s]                       # evaluators.py:31-35
            x = x / x.norm(2, 1, keepdim=True)
            assert np.abs(x.numpy() - ref[s]).max() < 5e-6
        cat = evaluators.extract_cnn_feature(m, imgs, True)    # for_eval=True: splits concatenated (resnet.py:122-124)
        assert cat.shape == (4, len(out) * 2048)
        with pytest.raises(NotImplementedError):
            evaluators.extract_cnn_feature(m, imgs, False, modules=["layer4"])


# ------------------------------------------------------------------ 8f-4: input transform + TripletLoss pairwise block
def test_preprocess_oracle_matches_pil_golden(golden):
    """oracle/preprocess_oracle.py (restatement of Pillow's bilinear resampling + ToTensor + Normalize) against PIL's own
    outputs stored in tests/golden/preprocess.npz; the host-side coefficient tables of the product are the same integers."""
    from oracle import preprocess_oracle as po
    from ssg_amd import preprocessor
    g = golden("preprocess.npz")
    for name in ("market", "duke", "up", "same", "split384"):
        H, W = (int(v) for v in g["size_" + name])
        for img, res in zip(g["in_" + name], g["resized_" + name]):
            assert np.array_equal(po.resize_bilinear_u8(img, H, W), res)
        x = po.to_tensor_normalize(g["resized_" + name][0])
        assert x.dtype == np.float32 and x.shape == (3, H, W)
        assert np.array_equal(x[1, 5, 7], (np.float32(g["resized_" + name][0][5, 7, 1]) / np.float32(255) - np.float32(0.456)) / np.float32(0.224))
        for n_in, n_out in ((g["in_" + name].shape[2], W), (g["in_" + name].shape[1], H)):
            a = po.bilinear_coeffs(n_in, n_out); b = preprocessor.bilinear_coeffs(n_in, n_out)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_gpu_preprocess_bit_exact_with_pil(golden, tmp_path):
    """ssg_preprocess_u8 == Resize((H,W)) + ToTensor + Normalize of the reference loaders (selftraining.py:43-47), bit for bit,
    on PIL's own outputs; then the loader surface on real files (decode on the CPU like preprocessor.py:28, transform on the GPU)."""
    from PIL import Image
    from oracle import preprocess_oracle as po
    from ssg_amd import preprocessor
    g = golden("preprocess.npz")
    expect = {}
    for name in ("market", "duke", "up", "same", "split384"):
        H, W = (int(v) for v in g["size_" + name])
        expect[name] = np.stack([po.to_tensor_normalize(r) for r in g["resized_" + name]])     # ToTensor + Normalize of PIL's own resize
        got = preprocessor.preprocess_batch(g["in_" + name], H, W).cpu().numpy()
        assert got.shape == expect[name].shape and got.dtype == np.float32
        assert np.array_equal(got, expect[name]), name
    # files of two different sizes through the DataLoader replacement, batch of 4, dataset order kept
    ds = []
    for i, name in enumerate(["market", "duke", "market", "duke", "market"]):
        fn = "img%d.png" % i
        Image.fromarray(g["in_" + name][i % g["in_" + name].shape[0]]).save(str(tmp_path / fn))
        ds.append((fn, i + 10, i % 3))
    loader = preprocessor.GpuBatchLoader(ds, root=str(tmp_path), height=256, width=128, batch_size=4)
    batches = list(loader)
    assert len(loader) == 2 and [len(b[1]) for b in batches] == [4, 1]
    imgs = torch.cat([b[0] for b in batches]).cpu().numpy()
    assert [f for b in batches for f in b[1]] == [d[0] for d in ds] and [p for b in batches for p in b[2]] == [d[1] for d in ds]
    for i, name in enumerate(["market", "duke", "market", "duke", "market"]):
        assert np.array_equal(imgs[i], expect[name][i % g["in_" + name].shape[0]]), i
    item = preprocessor.Preprocessor(ds, root=str(tmp_path))[1]
    assert item[0].dtype == np.uint8 and item[0].shape == (210, 77, 3) and item[1:] == ("img1.png", 11, 1)


@pytest.mark.gpu
def test_triplet_pairwise_block_vs_torch():
    """reid/loss/triplet.py:28-31 (pairwise distance of a training batch) on the fp32-MFMA Gram kernel; float32 tolerance
    2e-5 relative to the largest distance (GEMM accumulation order), the clamp floor reproduced exactly on the diagonal."""
    from ssg_amd import triplet
    g = torch.Generator().manual_seed(5)
    for n, d in ((128, 2048), (96, 512), (50, 100)):
        x = torch.randn(n, d, generator=g)
        x[3] = x[1]
        dist = torch.pow(x, 2).sum(dim=1, keepdim=True).expand(n, n)
        dist = dist + dist.t()
        dist = dist.addmm(x, x.t(), beta=1, alpha=-2)
        ref = dist.clamp(min=1e-12).sqrt()
        got = triplet.pairwise_dist(x).cpu()
        assert got.shape == (n, n)
        off = ~torch.eye(n, dtype=torch.bool); off[1, 3] = off[3, 1] = False
        assert (got[off] - ref[off]).abs().max() < 2e-5 * ref.max()
        assert float(got.min()) >= 1e-6 - 1e-12          # sqrt(1e-12): the clamp floor, never NaN


@pytest.mark.gpu
def test_triplet_pairwise_block_backward_vs_torch():
    """`triplet.pairwise_dist` is differentiable, so it can stand where reid/loss/triplet.py:28-31 stands inside the
    TripletLoss of the fine-tune phase: the gradient w.r.t. the features (two HIP elementwise kernels around one fp32-MFMA GEMM) against
    torch.autograd through the reference's four lines -- with the hardest-positive / hardest-negative mining and MarginRankingLoss of
    reid/loss/triplet.py:32-45 as the loss, and with a dense random upstream gradient; duplicate rows (clamped distance: no gradient)."""
    from ssg_amd import triplet

    def ref_dist(x):
        n = x.shape[0]
        dist = torch.pow(x, 2).sum(dim=1, keepdim=True).expand(n, n)
        dist = dist + dist.t()
        dist = dist.addmm(x, x.t(), beta=1, alpha=-2)
        return dist.clamp(min=1e-12).sqrt()

    def triplet_loss(dist, targets, margin=0.5):
        n = dist.shape[0]
        mask = targets.expand(n, n).eq(targets.expand(n, n).t())
        ap = torch.stack([dist[i][mask[i]].max() for i in range(n)]); an = torch.stack([dist[i][mask[i] == 0].min() for i in range(n)])
        return torch.nn.functional.margin_ranking_loss(an, ap, torch.ones_like(an), margin=margin)
    g = torch.Generator().manual_seed(11)
    for n, d in ((64, 2048), (96, 500), (30, 37)):
        x0 = torch.randn(n, d, generator=g) * 0.3
        x0[5] = x0[2]                                   # an exact duplicate: sq = 0 < 1e-12, the clamp blocks the gradient of that pair
        targets = torch.arange(n) // 4
        up = torch.randn(n, n, generator=g)
        for mode in ("dense", "loss"):
            xr = x0.clone().double().requires_grad_(True)       # float64 reference
            dr = ref_dist(xr)
            (dr * up.double()).sum().backward() if mode == "dense" else triplet_loss(dr, targets).backward()
            xg = x0.clone().cuda().requires_grad_(True)
            dg = triplet.pairwise_dist(xg)
            assert dg.requires_grad and dg.is_cuda
            (dg * up.cuda()).sum().backward() if mode == "dense" else triplet_loss(dg, targets.cuda()).backward()
            got, ref = xg.grad.cpu().double(), xr.grad
            assert got.shape == ref.shape and torch.isfinite(got).all()
            scale = float(ref.abs().max())
            # rows 2 and 5 are exact duplicates: in float64 their squared distance is exactly 0 and the clamp blocks the pair's gradient; in
            # float32 (this kernel, and the reference's own float32 addmm alike) |x|^2 + |y|^2 - 2 x.y leaves a rounding residue instead of 0,
            # the pair's weight 1 / dist is huge and multiplies x_2 - x_5 = 0 through a cancellation: those two rows get a looser bound
            rest = torch.ones(n, dtype=torch.bool); rest[2] = rest[5] = False
            assert float((got - ref)[rest].abs().max()) < 3e-5 * max(scale, 1e-3), (n, d, mode, float((got - ref)[rest].abs().max()), scale)
            assert float((got - ref)[~rest].abs().max()) < 2e-3 * max(scale, 1e-3), (n, d, mode, float((got - ref)[~rest].abs().max()), scale)
    # CPU input tensors get their gradient back on the CPU
    xc = x0.clone().requires_grad_(True)
    triplet.pairwise_dist(xc).sum().backward()
    assert xc.grad is not None and xc.grad.device.type == "cpu"


@pytest.mark.gpu
def test_triplet_block_inside_the_references_loss_vs_golden(golden):
    """tests/golden/triplet_ref.npz = the reference's own TripletLoss (reid/loss/triplet.py:11-77, float64, both mining modes) on seeded
    batches: loss, precision, d loss / d features.  Here `ssg_amd.triplet.pairwise_dist` stands where lines :28-31 stand, the mining and
    MarginRankingLoss of :32-77 follow in torch -- the loss, the precision and the gradient that reaches the features agree with the
    reference's numbers to float32 accuracy."""
    from ssg_amd import triplet
    g = golden("triplet_ref.npz")

    def mine(dist, targets, semi, K=4):
        n = dist.shape[0]
        mask = targets.expand(n, n).eq(targets.expand(n, n).t())
        ap, an = [], []
        if semi:                                                    # triplet.py:48-55
            for i in range(n // K):
                for j in range(K):
                    row = i * K + j
                    neg = dist[row][mask[row] == 0].min().view(1)
                    for pair in range(j + 1, K):
                        ap.append(dist[row][i * K + pair].view(1)); an.append(neg)
        else:                                                       # triplet.py:57-61
            for i in range(n):
                ap.append(dist[i][mask[i]].max().view(1)); an.append(dist[i][mask[i] == 0].min().view(1))
        ap, an = torch.cat(ap), torch.cat(an)
        loss = torch.nn.functional.margin_ranking_loss(an, ap, torch.ones_like(an), margin=0.5)
        return loss, float((an.detach() > ap.detach()).sum()) / an.shape[0]

    for ci in range(int(g["cases"])):
        x = torch.from_numpy(g["x_%d" % ci]); targets = torch.from_numpy(g["targets_%d" % ci]).cuda()
        for semi in (True, False):
            tag = "%d_%s" % (ci, "semi" if semi else "ohem")
            xg = x.clone().cuda().requires_grad_(True)
            dist = triplet.pairwise_dist(xg)
            loss, prec = mine(dist, targets, semi)
            loss.backward()
            ref = torch.from_numpy(g["grad_" + tag]).double()
            # float32 distances of magnitude max(dist) carry ~1e-7 * max(dist) each; the loss is a mean of differences of two of them
            lerr = abs(float(loss.detach()) - float(g["loss_" + tag]))
            assert lerr < 1e-6 * max(1.0, float(dist.detach().max())), (tag, lerr, float(dist.detach().max()))
            assert abs(prec - float(g["prec_" + tag])) < 1e-7, tag              # (the reference holds it as a float32 tensor)
            err = float((xg.grad.cpu().double() - ref).abs().max())
            assert err < 3e-5 * max(float(ref.abs().max()), 1e-3), (tag, err)


# ------------------------------------------------------------------ fused embedding kernels: shape gating (host functions, CPU)
def test_fused_kernel_shape_gates():
    """ssg_stem_pool_supported / ssg_bottleneck_supported decide between the fused kernels and the launch-per-layer path; the
    Python embedder relies on them (resnet._fmap / _bottleneck), so unsupported shapes must say 0 instead of being mis-run."""
    from ssg_amd import _lib
    L = _lib.lib()
    assert L.ssg_stem_pool_supported(256, 128) == 1 and L.ssg_stem_pool_supported(384, 128) == 1
    assert L.ssg_stem_pool_supported(256, 64) == 0 and L.ssg_stem_pool_supported(254, 128) == 0 and L.ssg_stem_pool_supported(4, 128) == 0
    ok = {(64, 32, 256, 256, 64), (64, 32, 64, 256, 64), (24, 32, 256, 256, 64), (32, 16, 512, 512, 128), (96, 16, 512, 512, 128)}
    for shape in ok:
        assert L.ssg_bottleneck_supported(*shape) == 1, shape
    bad = [(26, 32, 256, 256, 64), (64, 16, 256, 256, 64), (64, 32, 128, 256, 64), (28, 16, 512, 512, 128), (32, 16, 256, 512, 128),
           (16, 8, 1024, 1024, 256), (8, 4, 2048, 2048, 512), (0, 32, 256, 256, 64), (64, 32, 256, 256, 128)]
    for shape in bad:
        assert L.ssg_bottleneck_supported(*shape) == 0, shape
    # the entry points refuse what the gates refuse (before touching any pointer)
    for fn, args in ((L.ssg_bottleneck_nhwc_x, (None,) * 11 + (1, 16, 8, 1024, 256, None, None)),
                     (L.ssg_bottleneck_ds_nhwc_x, (None,) * 11 + (1, 64, 32, 256, 256, 64, None, None)),
                     (L.ssg_stem_pool_nchw_x, (None, 0, None, None, None, None, 1, 256, 64, None, None))):
        assert fn(*args) != 0 and b"unsupported" in L.ssg_last_error()


def test_bench_quotes_pmc_traffic_only_for_its_own_build(tmp_path):
    """bench.py's `roofline.traffic` comes from the PMC summary under profiles/; the summary carries the fingerprint of the kernel sources
    it was measured on and is refused for any other build, launch set or batch size (a stale byte count must not be quoted)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ssg_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fp = bench.build_fingerprint()
    assert len(fp) == 16 and fp == bench.build_fingerprint() and int(fp, 16) >= 0
    rec = {"build": fp, "batch": 1000, "launches_per_forward": 37, "fetch_bytes_per_forward": 35.2e9, "write_bytes_per_forward": 18.8e9,
           "algorithmic_bytes_per_forward": 41.9e9, "source": "unit test"}
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps(rec))
    got, note = bench.pmc_traffic(str(path), fp, 1000, 37)
    assert got == round((35.2e9 + 18.8e9) / 37) and "37 launches" in note
    assert bench.pmc_traffic(str(path), "0" * 16, 1000, 37)[0] is None and "another build" in bench.pmc_traffic(str(path), "0" * 16, 1000, 37)[1]
    assert bench.pmc_traffic(str(path), fp, 512, 37)[0] is None and bench.pmc_traffic(str(path), fp, 1000, 35)[0] is None
    assert bench.pmc_traffic(str(tmp_path / "missing.json"), fp, 1000, 37) == (None, None)
    path.write_text("{not json")
    assert bench.pmc_traffic(str(path), fp, 1000, 37) == (None, None)
    # the committed summary belongs to the committed kernels
    # the committed summary (when this round has one) belongs to the committed embedding kernels; bench.py refuses a stale one by itself
    if os.path.exists(bench.PMC_TRAFFIC_JSON):
        assert json.load(open(bench.PMC_TRAFFIC_JSON))["embed_build"] == bench.embed_fingerprint(), \
            "%s is stale: re-run tools/pmc_embed.sh on the GPU box" % bench.PMC_TRAFFIC_JSON
