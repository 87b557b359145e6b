"""CPU suite: the torch-fp32 restatement of the embedding path against the golden features
produced by the real reference model (tools/make_golden.py, tests/golden/embed_ref.npz)."""
import numpy as np
import torch


def test_embed_restatement_matches_reference_golden(golden):
    import ssg_amd
    from oracle import embed_oracle
    g = golden("embed_ref.npz")
    sd = ssg_amd.synthetic_state_dict(seed=int(g["weight_seed"]))
    imgs = torch.randn(4, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    for S in (2, 1):
        mine = torch.stack(embed_oracle.embed_with_flip(sd, imgs[:2], S)).numpy()
        ref = g["feats_S%d" % S][:, :2]
        assert mine.shape == ref.shape
        # same machine/threads -> identical; other hosts may pick other oneDNN kernels
        assert np.abs(mine - ref).max() < 2e-6
        assert np.allclose(np.linalg.norm(mine, axis=2), 1.0, atol=1e-5)


def test_embed_restatement_matches_the_wide_reference_goldens(golden):
    """round 5: the 16-image fixture and the checkpoint-like-BatchNorm fixture, both written by the REAL reference model
    (tools/make_golden.py --only-embed-wide); the restatement is checked on the first two images of each (CPU time)."""
    import ssg_amd
    from oracle import embed_oracle
    from synth import checkpoint_like_state_dict
    for fname, mk in (("embed_ref16.npz", lambda s: ssg_amd.synthetic_state_dict(seed=s)), ("embed_ckpt_ref.npz", checkpoint_like_state_dict)):
        g = golden(fname)
        n = int(g["n"])
        sd = mk(int(g["weight_seed"]))
        imgs = torch.randn(n, 3, 256, 128, generator=torch.Generator().manual_seed(int(g["image_seed"])))
        mine = torch.stack(embed_oracle.embed_with_flip(sd, imgs[:2], 2)).numpy()
        ref = g["feats_S2"]
        assert ref.shape == (3, n, 2048) and np.allclose(np.linalg.norm(ref, axis=2), 1.0, atol=1e-5)
        assert np.abs(mine - ref[:, :2]).max() < 2e-6, fname
    sd = checkpoint_like_state_dict(7)
    sc = (sd["base.layer2.0.bn2.weight"] / torch.sqrt(sd["base.layer2.0.bn2.running_var"] + 1e-5)).abs()
    assert float(sc.max() / sc.min()) > 1e3          # the point of the second fixture: per-channel scales spanning decades


def test_state_dict_surface():
    import ssg_amd
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False)
    sd = m.state_dict()
    assert "base.conv1.weight" in sd and "base.layer4.2.bn3.running_var" in sd and "feat.weight" in sd and "feat_bn.running_mean" in sd
    assert sd["base.layer2.0.downsample.0.weight"].shape == (512, 256, 1, 1)
    conv_params = sum(v.numel() for k, v in sd.items() if k.endswith("weight") and v.dim() == 4)
    assert conv_params == 23454912          # 23.5 M conv parameters
    sd2 = {k: v for k, v in sd.items() if not k.startswith("base.fc")}
    missing, unexpected = m.load_state_dict(sd2, strict=False)
    assert set(missing) == {"base.fc.weight", "base.fc.bias"} and not unexpected
    try:
        m(torch.zeros(1, 3, 256, 128))
        assert False, "CPU forward must not silently work"
    except ssg_amd.SSGError:
        pass
