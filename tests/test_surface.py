"""Small call-surface rows of SURVEY.md 8(a) that the big parity files do not exercise on their own:
a3 extract_cnn_feature, a2 fliplr, a10 the generate_dataloader label join, the DeviceBackedArray handle rules."""
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
