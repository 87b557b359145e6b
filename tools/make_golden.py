#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference (container-only).

/root/reference never travels to the GPU box; this script is run HERE, its outputs
(small .npz fixtures: inputs + expected outputs) are committed.  It also pins the CPU
oracle (oracle/ssg_oracle.c) against the reference while generating, and fails if the
oracle disagrees with the reference on any fixture.

Reference entry points exercised (paths relative to /root/reference):
  reid/rerank.py:27-127   re_ranking       (loaded by path; needs only numpy+scipy)
  selftraining.py:289-293 epsilon rule     (restated verbatim below for the per-stage fixtures; selftraining_fixture() runs the
                                            reference's own compute_dist / generate_selflabel / generate_dataloader, imported
                                            under the stub modules of import_reid(): selftraining.py needs torchvision at :14)
  selftraining.py:295,306 sklearn.cluster.DBSCAN(eps, min_samples=4, metric='precomputed')
Stage boundaries inside re_ranking (V, V_qe, jaccard) are captured with sys.settrace on the
reference frame -- no source edit, no copy.
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import ssg_oracle as ora  # noqa: E402
from sklearn.cluster import DBSCAN  # noqa: E402


def load_ref_rerank():
    spec = importlib.util.spec_from_file_location("ref_rerank", os.path.join(REF, "reid", "rerank.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.print = lambda *a, **k: None
    return m


class StableNp:
    """numpy proxy whose argsort is stable (pins tie order without editing the reference)."""

    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def argsort(a, *_, **__):
        return np.argsort(a, kind="stable")


def run_ref(mod, src, tgt, stable, **kw):
    """Run the reference re_ranking, capturing stage locals via a line tracer."""
    cap = {}
    code = mod.re_ranking.__code__

    def tracer(frame, event, arg):
        if frame.f_code is not code:
            return None
        if event == "line":
            ln = frame.f_lineno
            loc = frame.f_locals
            if ln == 94 and "V" in loc and "V" not in cap:          # after the k-reciprocal loop
                cap["V"] = loc["V"].copy(); cap["rank"] = loc["initial_rank"][:, : kw.get("k1", 20) + 1].copy()
                cap["Dn"] = loc["original_dist"].copy()
            if ln == 100 and "V" in loc:                              # after query expansion
                cap["V_qe"] = loc["V"].copy()
            if ln == 122 and "jaccard_dist" in loc:                   # after clamp, before fusion
                cap["jaccard"] = loc["jaccard_dist"].copy()
                cap["source_dist_row0"] = loc["source_dist"][0].copy()
        return tracer

    mod.np = StableNp() if stable else np
    sys.settrace(tracer)
    try:
        e, f = mod.re_ranking(src, tgt, **kw)
    finally:
        sys.settrace(None)
        mod.np = np
    return e, f, cap


def eps_rule_ref(dist, rho):
    # selftraining.py:289-293, verbatim semantics
    tri_mat = np.triu(dist, 1)
    tri_mat = tri_mat[np.nonzero(tri_mat)]
    tri_mat = np.sort(tri_mat, axis=None)
    top_num = np.round(rho * tri_mat.size).astype(int)
    eps = tri_mat[:top_num].mean()
    return eps, tri_mat.size, int(top_num)


def clustered(N, d, seed, per_id=16, intra=0.5):
    """Track-G synthetic embeddings: unit-norm identity centres + noise."""
    rng = np.random.default_rng(seed)
    P = max(1, N // per_id)
    c = rng.standard_normal((P, d)); c /= np.linalg.norm(c, axis=1, keepdims=True)
    sigma = np.sqrt(intra / 2.0 / d)
    ids = np.arange(N) % P
    x = c[ids] + sigma * rng.standard_normal((N, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def beq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.float16:
        a = a.view(np.uint16); b = b.view(np.uint16)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=(a.dtype.kind == "f"))


def exp_quirk_inputs():
    """Half inputs x where this host's numpy np.exp(half(-x)) is NOT the correctly rounded
    result (AVX512-FP16 SVML path).  Fixtures must not exercise them."""
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    with np.errstate(all="ignore"):
        npx = np.exp(allh)
    cr = ora.half_exp_table()
    bad = np.nonzero((npx.view(np.uint16) != cr.view(np.uint16)) & ~(np.isnan(npx) & np.isnan(cr)))[0]
    return allh, npx, cr, bad


def import_reid():
    """import the reference package under stub torchvision / h5py / metric_learn modules"""
    import types
    if "reid" in sys.modules:
        return sys.modules["reid"]
    spec = importlib.util.spec_from_file_location("ref_base", os.path.join(REF, "reid", "models", "base.py"))
    base = importlib.util.module_from_spec(spec); spec.loader.exec_module(base)
    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tvt = types.ModuleType("torchvision.transforms")
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, (lambda f: (lambda pretrained=False, **kw: f(pretrained=False, last_conv_stride=2)))(getattr(base, n)))
    names = ["Resize", "ToTensor", "Normalize", "Compose", "RandomHorizontalFlip", "TenCrop", "Lambda", "CenterCrop"]
    for n in names:
        setattr(tvt, n, type(n, (), {}))
    tvt.__all__ = names
    tv.models = tvm; tv.transforms = tvt
    h5 = types.ModuleType("h5py"); h5.File = object
    ml = types.ModuleType("metric_learn")
    for n in ("ITML_Supervised", "LMNN", "LSML_Supervised", "SDML_Supervised", "NCA", "LFDA", "RCA_Supervised"):
        setattr(ml, n, object)
    mlb = types.ModuleType("metric_learn.base_metric"); mlb.BaseMetricLearner = object
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt, "h5py": h5,
                        "metric_learn": ml, "metric_learn.base_metric": mlb})
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import reid
    return reid


def embed_fixture():
    """Embedding golden: the REAL reference model (reid.models.create + reid.evaluators.
    extract_features) under stub torchvision/h5py/metric_learn modules,
    loaded with the build's seeded synthetic weights, on 4 seeded 256x128 images."""
    import torch
    import ssg_amd
    from oracle import embed_oracle
    reid = import_reid()
    torch.manual_seed(0)
    ok = True
    rec = {}
    for S in (2, 1):
        model = reid.models.create('resnet50', num_classes=0, num_split=S, cluster=False)
        sd = ssg_amd.synthetic_state_dict(seed=1)
        missing = model.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys, missing
        model.eval()
        g = torch.Generator().manual_seed(1)
        imgs = torch.randn(4, 3, 256, 128, generator=g)
        loader = [(imgs, ["f%d" % i for i in range(4)], [0, 1, 2, 3], [0, 0, 0, 0])]
        feats, _ = reid.evaluators.extract_features(model, loader, for_eval=False)
        if S > 1:
            ref = torch.stack([torch.stack(feats["f%d" % i]) for i in range(4)], 1)      # [S+1, 4, 2048]
        else:
            ref = torch.stack([feats["f%d" % i] for i in range(4)]).unsqueeze(0)
        mine = torch.stack(embed_oracle.embed_with_flip(sd, imgs, S))
        err = (ref - mine).abs().max().item()
        print("embed S=%d: torch-restatement vs reference max|diff| = %.3e  (feature |max| %.3e)" % (S, err, ref.abs().max().item()))
        ok = ok and err < 1e-6
        rec["feats_S%d" % S] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, "embed_ref.npz"), image_seed=1, weight_seed=1, **rec)
    return ok


def embed_fixture_wide():
    """Two more embedding goldens from the REAL reference model:
      embed_ref16.npz     -- 16 seeded images (SURVEY.md 8c allows up to 16), seeded Kaiming weights, S = 2;
      embed_ckpt_ref.npz  -- 8 seeded images under checkpoint-like BatchNorm statistics (tools/synth.checkpoint_like_state_dict: folded
                             per-channel scales spanning > 10^3), S = 2: the split-half path pinned by the reference itself on realistic
                             weights, not only by the builder's restatement."""
    import torch
    import ssg_amd
    from oracle import embed_oracle
    from synth import checkpoint_like_state_dict
    reid = import_reid()
    ok = True
    for fname, sd, wseed, iseed, n, kind in (("embed_ref16.npz", ssg_amd.synthetic_state_dict(seed=1), 1, 16, 16, "kaiming"),
                                             ("embed_ckpt_ref.npz", checkpoint_like_state_dict(7), 7, 17, 8, "checkpoint-like")):
        torch.manual_seed(0)
        model = reid.models.create('resnet50', num_classes=0, num_split=2, cluster=False)
        missing = model.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys, missing
        model.eval()
        imgs = torch.randn(n, 3, 256, 128, generator=torch.Generator().manual_seed(iseed))
        names = ["f%d" % i for i in range(n)]
        loader = [(imgs[:5], names[:5], list(range(5)), [0] * 5), (imgs[5:], names[5:], list(range(5, n)), [0] * (n - 5))]
        feats, _ = reid.evaluators.extract_features(model, loader, for_eval=False)
        ref = torch.stack([torch.stack(feats[f]) for f in names], 1)                     # [3, n, 2048]
        mine = torch.stack(embed_oracle.embed_with_flip(sd, imgs, 2))
        err = (ref - mine).abs().max().item()
        sc = (sd["base.layer2.0.bn2.weight"] / torch.sqrt(sd["base.layer2.0.bn2.running_var"] + 1e-5)).abs()
        print("embed %s (%d images, %s weights, layer2.0.bn2 scale spread %.1e): torch-restatement vs reference max|diff| = %.3e"
              % (fname, n, kind, float(sc.max() / sc.min()), err))
        ok = ok and err < 2e-6
        np.savez_compressed(os.path.join(OUT, fname), image_seed=iseed, weight_seed=wseed, n=n, weights=kind, feats_S2=ref.numpy())
    return ok


def eval_fixture():
    """Retrieval metrics (reid/evaluators.py:88-129 evaluate_all -> reid/evaluation_metrics/ranking.py cmc, mean_ap with
    sklearn's average_precision_score): random query x gallery float32 distance blocks with Market-like id / camera
    structure (same-camera true matches that must be filtered, queries without any valid match, duplicate gallery
    vectors -> equal distances)."""
    import contextlib
    import io
    from oracle import eval_oracle
    reid = import_reid()
    from reid.evaluation_metrics import ranking
    import reid.evaluators as rev
    ok = True
    rec = {}
    for tag, m, n, nid, ncam, seed, dup in (("a", 40, 300, 25, 6, 41, False), ("b", 64, 500, 30, 2, 42, True), ("c", 16, 120, 40, 3, 43, False)):
        rng = np.random.default_rng(seed)
        gid = rng.integers(0, nid, n); gcam = rng.integers(0, ncam, n)
        qid = rng.integers(0, nid + (3 if tag == "c" else 0), m); qcam = rng.integers(0, ncam, m)   # case c: some ids absent from the gallery
        centres = rng.standard_normal((nid + 3, 16))
        gf = (centres[gid] + 1.6 * rng.standard_normal((n, 16))).astype(np.float32)
        qf = (centres[qid] + 1.6 * rng.standard_normal((m, 16))).astype(np.float32)
        if dup:
            gf[n // 2:n // 2 + 20] = gf[:20]          # duplicated gallery vectors: exactly equal distances
            gid[n // 2:n // 2 + 20] = gid[:20]        # (same identity, so cmc does not depend on the tie order)
        dist = ((qf[:, None, :] - gf[None, :, :]) ** 2).sum(-1).astype(np.float32)
        with contextlib.redirect_stdout(io.StringIO()):
            ref_map = ranking.mean_ap(dist, qid, gid, qcam, gcam)
            ref_cmc = ranking.cmc(dist, qid, gid, qcam, gcam, separate_camera_set=False, single_gallery_shot=False, first_match_break=True)
            ref_cmc_all = ranking.cmc(dist, qid, gid, qcam, gcam)                               # 'allshots' configuration
            ref_ret = rev.evaluate_all(dist, query_ids=qid, gallery_ids=gid, query_cams=qcam, gallery_cams=gcam)
        o_map = eval_oracle.mean_ap(dist, qid, gid, qcam, gcam)
        o_cmc = eval_oracle.cmc(dist, qid, gid, qcam, gcam, first_match_break=True)
        o_cmc_all = eval_oracle.cmc(dist, qid, gid, qcam, gcam)
        good = (o_map == ref_map) and np.array_equal(o_cmc, ref_cmc) and np.array_equal(o_cmc_all, ref_cmc_all) and ref_ret == ref_cmc[0]
        first, ap = eval_oracle.per_query(dist, qid, gid, qcam, gcam)
        print("eval %s: m=%d n=%d mAP=%.6f top1=%.4f valid queries=%d  oracle==reference (bitwise): %s" % (
            tag, m, n, ref_map, ref_cmc[0], int((first >= 0).sum()), good))
        ok = ok and bool(good)
        rec.update({"dist_" + tag: dist, "qid_" + tag: qid.astype(np.int32), "gid_" + tag: gid.astype(np.int32), "qcam_" + tag: qcam.astype(np.int32),
                    "gcam_" + tag: gcam.astype(np.int32), "map_" + tag: np.float64(ref_map), "cmc_" + tag: ref_cmc, "cmc_all_" + tag: ref_cmc_all,
                    "first_" + tag: first, "ap_" + tag: ap})
    np.savez_compressed(os.path.join(OUT, "eval_cases.npz"), **rec)
    return ok


def pairwise_fixture():
    """reid/evaluators.py:63-85 pairwise_distance (float32 torch on the CPU) called as shipped: the query x gallery branch
    (:74-85, deprecated positional addmm_) and the features-only branch (:64-72) on unit-norm and on un-normalised features."""
    import contextlib
    import io
    import warnings
    from collections import OrderedDict
    import torch
    reid = import_reid()
    import reid.evaluators as rev
    rec = {}
    for tag, n, d, unit, seed in (("u", 70, 200, True, 3), ("r", 45, 2048, False, 4)):
        g = torch.Generator().manual_seed(seed)
        feats = OrderedDict()
        for i in range(n):
            f = torch.randn(d, generator=g) * (1.0 if unit else 0.3)
            feats["f%03d" % i] = f / f.norm() if unit else f
        query = [("f%03d" % i, 0, 0) for i in range(0, n // 2)]
        gallery = [("f%03d" % i, 0, 0) for i in range(n // 3, n)]
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            qg = rev.pairwise_distance(feats, query, gallery).numpy()
            full = rev.pairwise_distance(feats).numpy()
        rec.update({"feats_" + tag: torch.stack(list(feats.values())).numpy(), "nq_" + tag: np.int32(n // 2), "g0_" + tag: np.int32(n // 3),
                    "qg_" + tag: qg, "self_" + tag: full})
        print("pairwise %s: n=%d d=%d  qg %r self %r (reference outputs stored)" % (tag, n, d, qg.shape, full.shape))
    np.savez_compressed(os.path.join(OUT, "pairwise.npz"), **rec)
    return True


def jpeg_fixture():
    """Decode half of the input side (reid/utils/data/preprocessor.py:28 `Image.open(fpath).convert('RGB')`): small JPEG files
    written with Pillow (the reference's own codec: baseline Huffman, 4:4:4 / 4:2:2 / 4:2:0, grayscale, restart markers,
    optimised Huffman tables, odd sizes, a Market-1501-sized 128 x 64 image) and the pixels Pillow decodes from them.  The
    fixture stores file bytes + expected RGB arrays; oracle/jpeg_oracle.py must reproduce every one of them."""
    import io
    from PIL import Image
    from oracle import jpeg_oracle
    rng = np.random.default_rng(77)
    rec, ok, n = {}, True, 0

    def picture(h, w, kind):
        if kind == 0:
            a = rng.integers(0, 256, (h, w, 3))
        elif kind == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([xx * 255.0 / max(w - 1, 1), yy * 255.0 / max(h - 1, 1), (xx + yy) * 127.0 / max(h + w - 2, 1)], -1) + rng.normal(0, 6, (h, w, 3))
        else:
            a = np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 3)), np.ones((8, 8, 1)))[:h, :w] + rng.normal(0, 20, (h, w, 3))
        return np.clip(a, 0, 255).astype(np.uint8)
    cases = [(128, 64, 1, 75, 2, 0, False, False), (128, 64, 2, 90, 2, 0, False, False), (37, 53, 0, 60, 2, 0, False, False), (17, 9, 1, 85, 1, 0, False, False),
             (64, 33, 2, 95, 0, 0, False, False), (31, 64, 1, 40, 2, 3, False, False), (100, 7, 2, 75, 1, 2, False, False), (5, 3, 0, 75, 2, 0, False, False),
             (48, 40, 1, 80, 2, 0, True, False), (40, 24, 1, 85, 0, 0, False, True), (1, 1, 0, 75, 2, 0, False, False), (200, 100, 2, 70, 2, 5, True, False)]
    for h, w, kind, q, ss, ri, opt, gray in cases:
        a = picture(h, w, kind)
        if gray:
            a = a[:, :, 0]
        kw = dict(quality=q, optimize=opt)
        if not gray:
            kw["subsampling"] = ss
        if ri:
            kw["restart_marker_blocks"] = ri
        buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        good = np.array_equal(jpeg_oracle.decode(data), ref)
        ok = ok and good
        rec["file_%02d" % n] = np.frombuffer(data, np.uint8); rec["rgb_%02d" % n] = ref
        print("jpeg %02d: %dx%d q=%d subsampling=%d restart=%d optimize=%s gray=%s  %d bytes  oracle==Pillow: %s" % (n, h, w, q, ss, ri, opt, gray, len(data), good))
        n += 1
    # a progressive file: outside the GPU decoder's class (the product leaves it to Pillow); the expected pixels are still Pillow's
    buf = io.BytesIO(); Image.fromarray(picture(40, 40, 1)).save(buf, "JPEG", quality=80, progressive=True)
    rec["progressive_file"] = np.frombuffer(buf.getvalue(), np.uint8); rec["progressive_rgb"] = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    rec["count"] = np.int32(n)
    np.savez_compressed(os.path.join(OUT, "jpeg_cases.npz"), **rec)
    return ok


def plain_fixture():
    """reid/rerank_plain.py re_ranking (kNN-set Jaccard variant): oracle restatement vs the reference."""
    import contextlib
    import io
    spec = importlib.util.spec_from_file_location("ref_plain", os.path.join(REF, "reid", "rerank_plain.py"))
    mp = importlib.util.module_from_spec(spec); spec.loader.exec_module(mp)
    ok = True
    rec = {}
    for tag, N, Ns, d, seed, lam, k in (("a", 96, 64, 32, 51, 0.1, 20), ("b", 300, 200, 64, 52, 0.3, 12), ("c", 512, 384, 64, 53, 0.1, 20)):
        tgt = clustered(N, d, seed); src = clustered(Ns, d, seed + 1000, intra=0.6)
        if tag == "b":
            tgt[5] = tgt[9]                                  # duplicate rows: ties at distance 0
        with contextlib.redirect_stdout(io.StringIO()):
            ref, ref2 = mp.re_ranking(src, tgt, k=k, lambda_value=lam)
        mine, _, st = ora.re_ranking_plain(src, tgt, k=k, lambda_value=lam, stages=True)
        good = beq(ref, mine) and ref is ref2
        sizes = st["knn"].sum(axis=1)
        print("rerank_plain %s: N=%d k=%d lambda=%.1f set sizes %d..%d  oracle==reference (bitwise): %s" % (tag, N, k, lam, sizes.min(), sizes.max(), good))
        ok = ok and bool(good)
        rho = 2e-2 if N < 256 else 1.6e-3
        eps, cnt, top = eps_rule_ref(ref, rho)
        labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(ref)
        rec.update({"src_" + tag: src, "tgt_" + tag: tgt, "k_" + tag: k, "lam_" + tag: lam, "final_" + tag: ref, "rho_" + tag: rho,
                    "eps_" + tag: np.float64(eps), "labels_" + tag: labels.astype(np.int64), "setsize_" + tag: sizes.astype(np.int32)})
    np.savez_compressed(os.path.join(OUT, "rerank_plain.npz"), **rec)
    return ok


def init_fixture(mod):
    """re_ranking_init (float32 cosine variant, rerank.py:171-234 / rerank_initial.py:40-99)."""
    spec = importlib.util.spec_from_file_location("ref_init", os.path.join(REF, "reid", "rerank_initial.py"))
    m2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(m2)
    ok = True
    rec = {}
    for tag, nq, ng, d, seed, lam, k1, k2 in (("a", 64, 192, 64, 31, 0.3, 20, 6), ("b", 48, 300, 96, 32, 0.1, 10, 4)):
        x = clustered(nq + ng, d, seed)
        q, g = x[:nq], x[nq:]
        ref = mod.re_ranking_init(q, g, k1=k1, k2=k2, lambda_value=lam)
        ref2 = m2.re_ranking_init(np.dot(q, g.T), np.dot(q, q.T), np.dot(g, g.T), k1=k1, k2=k2, lambda_value=lam)
        mine = ora.re_ranking_init(q, g, k1=k1, k2=k2, lambda_value=lam)
        err = float(np.abs(mine - ref).max())
        print("re_ranking_init %s: both reference copies identical: %s; oracle max|diff| = %.2e" % (tag, np.array_equal(ref, ref2), err))
        ok = ok and np.array_equal(ref, ref2) and err < 2e-6
        rec.update({"q_" + tag: q, "g_" + tag: g, "final_" + tag: ref, "k1_" + tag: k1, "k2_" + tag: k2, "lam_" + tag: lam})
    np.savez_compressed(os.path.join(OUT, "rerank_init.npz"), **rec)
    return ok


def tiefree_fixture(mod):
    """A fixture from the UNTOUCHED reference on an input whose normalised half distances are distinct inside the
    first k1+2 columns of every row (SURVEY.md 7 hard part 1a): np.argsort's tie behaviour cannot matter, so every
    rank mode of the build must reproduce it.  Seeds are searched until the input is tie free."""
    lam, rho = 0.3, 2e-2
    for N, Ns, d, k1, k2 in ((40, 32, 48, 6, 3), (48, 40, 64, 4, 2)):
        for seed in range(1000, 6000):
            rng = np.random.default_rng(seed)
            tgt = rng.standard_normal((N, d)); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
            src = rng.standard_normal((Ns, d)); src /= np.linalg.norm(src, axis=1, keepdims=True)
            tgt = tgt.astype(np.float32); src = (0.5 * src + 0.5 * tgt[:Ns]).astype(np.float32)
            f16 = tgt.astype(np.float16).astype(np.float64)
            from scipy.spatial.distance import cdist
            D = np.power(cdist(f16, f16).astype(np.float16), 2).astype(np.float16)
            Dn = np.transpose(D / np.max(D, axis=0))
            if np.all(np.diff(np.sort(Dn.astype(np.float32), axis=1)[:, : k1 + 2], axis=1) != 0):
                break
        else:
            raise SystemExit("no tie-free seed found")
        e, f, cap = run_ref(mod, src, tgt, False, k1=k1, k2=k2, lambda_value=lam)
        e2, f2, cap2 = run_ref(mod, src, tgt, True, k1=k1, k2=k2, lambda_value=lam)
        assert beq(f, f2) and beq(cap["rank"], cap2["rank"]), "tie-free input must not depend on the argsort kind"
        ok = True
        for mode in ("introsort", "stable"):
            oe, of, st = ora.re_ranking(src, tgt, k1=k1, k2=k2, lambda_value=lam, rank_mode=mode, stages=True)
            chk = dict(euclid=beq(e, oe), rank=beq(cap["rank"], st["rank"]), V=beq(cap["V"], st["V"]), V_qe=beq(cap["V_qe"], st["V_qe"]),
                       jaccard=beq(cap["jaccard"], st["jaccard"]), final=beq(f, of))
            ok = ok and all(chk.values())
            print("tiefree N=%d seed=%d oracle(%s)==reference: %s" % (N, seed, mode, chk))
        eps, cnt, top = eps_rule_ref(f, rho)
        labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(f)
        ok = ok and beq(labels, ora.dbscan(f, eps, 4)) and (float(eps), cnt, top) == ora.eps_rule(f, rho)
        np.savez_compressed(os.path.join(OUT, "rerank_tiefree_n%d_ref.npz" % N), src=src, tgt=tgt, k1=k1, k2=k2, lambda_value=lam, rho=rho,
                            stable=False, rank=cap["rank"].astype(np.int32), eps=np.float64(eps), count=cnt, top_num=top,
                            labels=labels.astype(np.int64), tie_free=True, exp_quirk=False, v=cap["source_dist_row0"].astype(np.float64),
                            euclid=e, final=f, V=cap["V"], V_qe=cap["V_qe"], jaccard=cap["jaccard"], seed=seed)
        if not ok:
            return False
    return True


def variant_fixtures(mod):
    """Untouched reference on the argument combinations outside the script defaults: MemorySave=True (rerank.py:49-59: the
    original distance is squared in float64 and rounded once; Minibatch chunks the rows) and k2 > k1+1 (the query expansion
    :97 then reads initial_rank columns beyond k1+1)."""
    ok = True
    for name, N, Ns, d, kw in (("msave", 96, 64, 64, dict(k1=20, k2=6, lambda_value=0.1, MemorySave=True, Minibatch=40)),
                               ("k2wide", 96, 64, 64, dict(k1=4, k2=9, lambda_value=0.3))):
        tgt = clustered(N, d, 31); src = clustered(Ns, d, 1031, intra=0.6)
        e, f, cap = run_ref(mod, src, tgt, False, **kw)
        oe, of, st = ora.re_ranking(src, tgt, rank_mode="introsort", stages=True, **kw)
        K = cap["rank"].shape[1]
        chk = dict(euclid=beq(e, oe), rank=beq(cap["rank"], st["rank"][:, :K]), V=beq(cap["V"], st["V"]), V_qe=beq(cap["V_qe"], st["V_qe"]),
                   jaccard=beq(cap["jaccard"], st["jaccard"]), final=beq(f, of))
        rho = 2e-2
        eps, cnt, top = eps_rule_ref(f, rho)
        labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(f)
        chk.update(eps=(float(eps), cnt, top) == ora.eps_rule(f, rho), labels=beq(labels, ora.dbscan(f, eps, 4)))
        if name == "msave":     # the two branches must really differ on this input, or the fixture pins nothing
            e0, _ = mod.re_ranking(src, tgt, k1=20, k2=6, lambda_value=0.1, no_rerank=True)
            assert not beq(e, e0), "MemorySave fixture does not distinguish the two rounding orders"
        print("variant %-7s oracle==reference: %s" % (name, chk))
        ok = ok and all(chk.values())
        np.savez_compressed(os.path.join(OUT, "rerank_var_%s_ref.npz" % name), src=src, tgt=tgt, k1=kw["k1"], k2=kw["k2"],
                            lambda_value=kw["lambda_value"], memory_save=bool(kw.get("MemorySave", False)), rho=rho, stable=False,
                            rank=cap["rank"].astype(np.int32), eps=np.float64(eps), count=cnt, top_num=top, labels=labels.astype(np.int64),
                            tie_free=False, exp_quirk=False, v=cap["source_dist_row0"].astype(np.float64), euclid=e, final=f, V=cap["V"],
                            V_qe=cap["V_qe"], jaccard=cap["jaccard"])
    return ok


def wide_fixture(mod):
    """the UNTOUCHED reference at the bench's feature width -- N = 2000, Ns = 2000, d = 2048 on the hard set
    (tools/synth.hard_clustered, lambda = 0.3): the 2048-term float64 cdist sums (rerank.py:37,61), the introsort tie order at
    N = 2000, eps and sklearn's labels.  The inputs are regenerated from their seeds by the tests (their sha256 is stored);
    the N x N outputs are stored as sha256 + the first 21 rank columns + eps + labels (small)."""
    import synth
    N, Ns, d, lam, rho = 2000, 2000, 2048, 0.3, 1.6e-3
    tgt = synth.hard_clustered(N, d, 31); src = synth.hard_clustered(Ns, d, 32, intra=0.7)
    e, f, cap = run_ref(mod, src, tgt, False, k1=20, k2=6, lambda_value=lam)
    ora.set_num_threads(8)
    oe, of, st = ora.re_ranking(src, tgt, k1=20, k2=6, lambda_value=lam, rank_mode="introsort", stages=True)
    chk = dict(euclid=beq(e, oe), rank=beq(cap["rank"], st["rank"]), V=beq(cap["V"], st["V"]), V_qe=beq(cap["V_qe"], st["V_qe"]),
               jaccard=beq(cap["jaccard"], st["jaccard"]), final=beq(f, of))
    eps, cnt, top = eps_rule_ref(f, rho)
    oeps, ocnt, otop = ora.eps_rule(f, rho)
    labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(f)
    chk.update(eps=(float(eps) == oeps and cnt == ocnt and top == otop), labels=beq(labels, ora.dbscan(f, eps, 4)))
    allh, npx, cr, bad = exp_quirk_inputs()
    used = set(np.unique((-cap["Dn"][cap["V"] != 0]).view(np.uint16)).tolist())
    quirky = bool(used & set(int(b) for b in bad))
    print("wide d=2048 N=%d Ns=%d oracle==reference: %s exp-quirk=%s eps=%.6f clusters=%d noise=%d" % (
        N, Ns, chk, quirky, eps, labels.max() + 1, int((labels < 0).sum())))
    # source vector exactly as the reference forms it (rerank.py:36-40) = row 0 of the captured source_dist
    np.savez_compressed(os.path.join(OUT, "rerank_wide_d2048_ref.npz"), N=N, Ns=Ns, d=d, seed_tgt=31, seed_src=32, intra_src=0.7,
                        lambda_value=lam, rho=rho, k1=20, k2=6, sha_tgt=sha(tgt), sha_src=sha(src), sha_euclid=sha(e), sha_final=sha(f),
                        sha_V=sha(cap["V"]), sha_Vqe=sha(cap["V_qe"]), sha_jaccard=sha(cap["jaccard"]),
                        rank=cap["rank"].astype(np.int32), v=cap["source_dist_row0"].astype(np.float64),
                        euclid_row0=e[0].copy(), final_row0=f[0].copy(), final_diag=np.diag(f).copy(),
                        eps=np.float64(eps), count=cnt, top_num=top, labels=labels.astype(np.int64), exp_quirk=quirky)
    return all(chk.values()) and not quirky


def selftraining_fixture():
    """a6 / a8 / a9 / a10 through the reference's OWN functions: /root/reference/selftraining.py imported by path under the stub modules of
    import_reid() (it needs torchvision at :14), then compute_dist (:255-277), generate_selflabel (:280-313; iteration 0 = eps rule +
    cached estimators, iteration 1 = the cached estimators on new distances, eps frozen) and generate_dataloader (:315-331; the
    DataLoader / Preprocessor / RandomIdentitySampler it wraps around the joined dataset are replaced by recorders through the
    module's globals -- no source edit) on 3 splits of synthetic embeddings.  The oracle is checked against every output."""
    import types
    import torch
    import_reid()
    spec = importlib.util.spec_from_file_location("ref_selftraining", os.path.join(REF, "selftraining.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.print = lambda *a, **k: None
    sys.modules["reid.rerank"].print = lambda *a, **k: None
    seen = {}
    m.Preprocessor = lambda dataset, root=None, transform=None: seen.setdefault("dataset", list(dataset))
    m.RandomIdentitySampler = lambda dataset, num_instances: None
    m.DataLoader = lambda ds, **kw: "loader"
    N, Ns, d, lam, rho, S1 = 200, 150, 64, 0.1, 6e-2, 3
    args = types.SimpleNamespace(no_rerank=False, rho=rho, batch_size=32, num_instances=4)
    rec = dict(N=N, Ns=Ns, d=d, lambda_value=lam, rho=rho, splits=S1)
    ok = True
    cluster_list = []
    for it in range(2):
        tgt = [clustered(N, d, 300 + 10 * it + s, intra=0.35) for s in range(S1)]
        src = [clustered(Ns, d, 1300 + 10 * it + s, intra=0.6) for s in range(S1)]
        e_list, r_list = m.compute_dist([torch.from_numpy(x) for x in src], [torch.from_numpy(x) for x in tgt], lam, False)
        assert e_list == [[]] * S1 and len(r_list) == S1 and r_list[0].dtype == np.float64
        labels_list, cluster_list = m.generate_selflabel(e_list, r_list, it, args, cluster_list)
        for s in range(S1):
            rec["src_%d_%d" % (it, s)] = src[s]; rec["tgt_%d_%d" % (it, s)] = tgt[s]
            rec["labels_%d_%d" % (it, s)] = np.asarray(labels_list[s]).astype(np.int64)
            rec["sha_final_%d_%d" % (it, s)] = sha(r_list[s])
            _, of = ora.re_ranking(src[s], tgt[s], lambda_value=lam)
            good = beq(r_list[s], of)
            if it == 0:
                rec["eps_%d" % s] = np.float64(cluster_list[s].eps)
                oeps, _, _ = ora.eps_rule(of, rho)
                good = good and float(cluster_list[s].eps) == oeps
            good = good and beq(np.asarray(labels_list[s]).astype(np.int64), ora.dbscan(of, float(cluster_list[s].eps), 4))
            ok = ok and bool(good)
        trainval = [("img_%05d_c%d.jpg" % (i, i % 6), i // 16, i % 6) for i in range(N)]
        seen.clear()
        loader = m.generate_dataloader(types.SimpleNamespace(trainval=trainval, images_dir="/nowhere"), labels_list, None, it, args)
        assert loader == "loader"
        ds = seen["dataset"]
        rec["kept_%d" % it] = np.array([int(f[4:9]) for f, _, _ in ds], np.int64)
        rec["kept_labels_%d" % it] = np.array([[int(x) for x in lab] for _, lab, _ in ds], np.int64).reshape(len(ds), S1)
        assert all(c == 0 for _, _, c in ds)
        print("selftraining it=%d: ids per split %s, %d of %d images kept, oracle==reference: %s" % (
            it, [len(set(l.tolist())) - (1 if -1 in l else 0) for l in labels_list], len(ds), N, ok))
    assert len(cluster_list) == S1                      # iteration 1 reused the cached estimators (eps frozen)
    np.savez_compressed(os.path.join(OUT, "selftraining_ref.npz"), **rec)
    return ok


def triplet_fixture():
    """f4 (triplet block): the reference's OWN TripletLoss (reid/loss/triplet.py:11-77, both mining modes: `use_semi` and plain
    hardest-positive / hardest-negative) run in float64 on seeded batches -- loss, precision and the gradient with respect to the
    features.  The product replaces lines :28-31 of its forward (ssg_amd.triplet.pairwise_dist); the GPU test feeds that block into the
    same mining and compares with these numbers."""
    import warnings
    import torch
    import_reid()
    from reid.loss import TripletLoss
    rec = {}
    g = torch.Generator().manual_seed(21)
    cases = ((32, 2048), (96, 500), (32, 37))
    for ci, (n, d) in enumerate(cases):
        x = torch.randn(n, d, generator=g) * 0.3
        targets = torch.arange(n) // 4
        rec["x_%d" % ci] = x.numpy(); rec["targets_%d" % ci] = targets.numpy()
        for semi in (True, False):
            xr = x.clone().double().requires_grad_(True)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                loss, prec = TripletLoss(margin=0.5, num_instances=4, use_semi=semi)(xr, targets, 0)
                loss.backward()
            tag = "%d_%s" % (ci, "semi" if semi else "ohem")
            rec["loss_" + tag] = np.float64(loss.item()); rec["prec_" + tag] = np.float64(float(prec)); rec["grad_" + tag] = xr.grad.numpy().astype(np.float32)
            print("triplet n=%d d=%d %s: loss %.6f prec %.4f" % (n, d, "semi" if semi else "ohem", loss.item(), float(prec)))
    rec["cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "triplet_ref.npz"), **rec)
    return True


def preprocess_fixture():
    """tests/golden/preprocess.npz: decoded-image inputs and what the reference's extraction transform makes of them
    (selftraining.py:43-47 via reid/utils/data/preprocessor.py:22-30).  The resize is run with PIL itself (what
    torchvision's Resize calls for a PIL image); torchvision is absent here, so ToTensor / Normalize are their published
    float32 formulas.  Also pins oracle/preprocess_oracle.py against PIL on many random shapes."""
    from PIL import Image
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(77)
    ok = True
    for t in range(200):
        h = int(rng.integers(4, 320)); w = int(rng.integers(4, 220))
        H, W = [(256, 128), (384, 128), (64, 32), (h, w), (300, 310), (h, 128), (256, w)][t % 7]
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((W, H), Image.BILINEAR))
        ok = ok and np.array_equal(ref, po.resize_bilinear_u8(img, H, W))
    print("preprocess oracle == PIL.Image.resize(BILINEAR) on 200 random shapes:", ok)
    rec = {}
    for name, (h, w), (H, W), B in (("market", (128, 64), (256, 128), 3), ("duke", (210, 77), (256, 128), 2), ("up", (40, 30), (256, 128), 1),
                                    ("same", (256, 128), (256, 128), 1), ("split384", (173, 91), (384, 128), 1)):
        imgs = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        imgs[0] = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1).astype(np.uint8)   # smooth ramps
        res = np.stack([np.asarray(Image.fromarray(im).resize((W, H), Image.BILINEAR)) for im in imgs])
        rec["in_" + name] = imgs; rec["resized_" + name] = res      # PIL's output; ToTensor/Normalize are applied by the tests (published formulas)
        rec["size_" + name] = np.asarray([H, W])
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **rec)
    return ok


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-plain" in sys.argv:        # regenerate just tests/golden/rerank_plain.npz
        ok = plain_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-preproc" in sys.argv:      # regenerate just tests/golden/preprocess.npz
        ok = preprocess_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-jpeg" in sys.argv:         # regenerate just tests/golden/jpeg_cases.npz
        ok = jpeg_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-pairwise" in sys.argv:     # regenerate just tests/golden/pairwise.npz
        ok = pairwise_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-triplet" in sys.argv:      # regenerate just tests/golden/triplet_ref.npz
        ok = triplet_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-selftraining" in sys.argv:  # regenerate just tests/golden/selftraining_ref.npz
        ora.build(force=False)
        ok = selftraining_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-embed-wide" in sys.argv:   # regenerate just tests/golden/embed_ref16.npz and embed_ckpt_ref.npz
        ok = embed_fixture_wide()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-eval" in sys.argv:         # regenerate just tests/golden/eval_cases.npz
        ok = eval_fixture()
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    ora.build(force=True)
    mod = load_ref_rerank()
    if "--only-wide" in sys.argv:         # regenerate just tests/golden/rerank_wide_d2048_ref.npz
        ok = wide_fixture(mod)
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    if "--only-tiefree" in sys.argv:      # regenerate just tests/golden/rerank_tiefree_*.npz and rerank_var_*.npz
        ok = tiefree_fixture(mod)
        ok = variant_fixtures(mod) and ok
        print("ALL OK" if ok else "ORACLE MISMATCH")
        sys.exit(0 if ok else 1)
    ok = tiefree_fixture(mod)
    ok = variant_fixtures(mod) and ok
    ok = preprocess_fixture() and ok
    ok = selftraining_fixture() and ok
    ok = triplet_fixture() and ok

    # ---- half exp table of this host's numpy + the exceptions vs correct rounding
    allh, npx, cr, bad = exp_quirk_inputs()
    print("np.exp(half) differs from correctly-rounded on %d inputs:" % len(bad), [float(allh[i]) for i in bad])
    np.savez_compressed(os.path.join(OUT, "half_exp_table.npz"), numpy_exp_bits=npx.view(np.uint16),
                        quirk_input_bits=bad.astype(np.uint16))
    quirk_neg = set(int(b) for b in bad)   # bit patterns of the (negative) exp arguments

    # ---- numpy primitive semantics the oracle restates
    rng = np.random.default_rng(123)
    for _ in range(2000):
        n = int(rng.integers(1, 700))
        a32 = rng.random(n).astype(np.float32)
        assert np.sum(a32) == ora.pairwise_sum(a32), "pairwise f32"
        a64 = rng.random(n)
        assert np.sum(a64) == ora.pairwise_sum(a64), "pairwise f64"
    for _ in range(300):
        n = int(rng.integers(2, 3000))
        h = (rng.integers(0, 40, n) / 64.0).astype(np.float16)    # tie-heavy
        assert np.array_equal(np.argsort(h), ora.argsort_half(h)), "introsort restatement"
        h = rng.random(n).astype(np.float16)
        assert np.array_equal(np.argsort(h), ora.argsort_half(h)), "introsort restatement"
    print("numpy primitives (pairwise sum f32/f64, half introsort argsort): oracle == numpy")

    # ---- re_ranking fixtures
    cases = [
        # name, N, Ns, d, lambda, k1, k2, seed, store_full
        ("n64_l01", 64, 48, 64, 0.1, 20, 6, 11, True),
        ("n64_l03_k8", 64, 80, 64, 0.3, 8, 3, 12, True),
        ("n256_l01", 256, 192, 128, 0.1, 20, 6, 13, True),
        ("n256_l03", 256, 256, 128, 0.3, 20, 6, 14, True),
        ("n1024_l01", 1024, 768, 64, 0.1, 20, 6, 15, False),
        ("n1024_l03", 1024, 1024, 64, 0.3, 20, 6, 16, False),
    ]
    for name, N, Ns, d, lam, k1, k2, seed, full in cases:
        tgt = clustered(N, d, seed)
        src = clustered(Ns, d, seed + 1000, intra=0.6)
        for variant, stable in (("ref", False), ("stable", True)):
            e, f, cap = run_ref(mod, src, tgt, stable, k1=k1, k2=k2, lambda_value=lam)
            oe, of, st = ora.re_ranking(src, tgt, k1=k1, k2=k2, lambda_value=lam,
                                        rank_mode="stable" if stable else "introsort", stages=True)
            # which exp arguments did the reference evaluate?  (weights use -Dn[i, idx])
            used = set(np.unique((-cap["Dn"][cap["V"] != 0]).view(np.uint16)).tolist())
            quirky = bool(used & quirk_neg)
            chk = dict(euclid=beq(e, oe), rank=beq(cap["rank"], st["rank"]), V=beq(cap["V"], st["V"]),
                       V_qe=beq(cap["V_qe"], st["V_qe"]), jaccard=beq(cap["jaccard"], st["jaccard"]),
                       final=beq(f, of))
            eps, cnt, top = eps_rule_ref(f, 1.6e-3 if N >= 256 else 2e-2)
            rho = 1.6e-3 if N >= 256 else 2e-2
            oeps, ocnt, otop = ora.eps_rule(f, rho)
            labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(f)
            olabels = ora.dbscan(f, eps, 4)
            chk.update(eps=(float(eps) == oeps and cnt == ocnt and top == otop), labels=beq(labels, olabels))
            tiefree = bool(np.all(np.diff(np.sort(cap["Dn"].astype(np.float32), axis=1)[:, : k1 + 2], axis=1) != 0))
            print("%-12s %-6s N=%d oracle==reference: %s  tie-free=%s exp-quirk=%s eps=%.6f ids=%d" % (
                name, variant, N, chk, tiefree, quirky, eps, len(set(labels.tolist())) - (1 if -1 in labels else 0)))
            if not all(chk.values()) and not quirky:
                ok = False
            rec = dict(src=src, tgt=tgt, k1=k1, k2=k2, lambda_value=lam, rho=rho, stable=stable,
                       rank=cap["rank"].astype(np.int32), eps=np.float64(eps), count=cnt, top_num=top,
                       labels=labels.astype(np.int64), tie_free=tiefree, exp_quirk=quirky,
                       v=(cap["source_dist_row0"] ).astype(np.float64),
                       sha_euclid=sha(e), sha_final=sha(f), sha_V=sha(cap["V"]), sha_Vqe=sha(cap["V_qe"]),
                       sha_jaccard=sha(cap["jaccard"]))
            if full:
                rec.update(euclid=e, final=f, V=cap["V"], V_qe=cap["V_qe"], jaccard=cap["jaccard"])
            np.savez_compressed(os.path.join(OUT, "rerank_%s_%s.npz" % (name, variant)), **rec)

    # ---- no-rerank path: euclidean half matrix -> eps (half) -> DBSCAN  (BASELINE config 1/2)
    for name, N, d, seed, rho in (("n256", 256, 128, 21, 1.6e-2), ("n1024", 1024, 64, 22, 1.6e-3)):
        tgt = clustered(N, d, seed)
        e, _ = mod.re_ranking(tgt[:8], tgt, no_rerank=True)
        oe, _ = ora.re_ranking(tgt[:8], tgt, no_rerank=True)
        eps, cnt, top = eps_rule_ref(e, rho)
        oeps, ocnt, otop = ora.eps_rule(e, rho)
        labels = DBSCAN(eps=eps, min_samples=4, metric="precomputed", n_jobs=8).fit_predict(e)
        olabels = ora.dbscan(e.astype(np.float64), float(eps), 4)
        good = beq(e, oe) and np.float16(eps).view(np.uint16) == np.float16(oeps).view(np.uint16) and cnt == ocnt and beq(labels, olabels)
        print("norerank %-6s oracle==reference: %s eps=%s ids=%d" % (name, good, eps, len(set(labels.tolist())) - 1))
        ok = ok and bool(good)
        np.savez_compressed(os.path.join(OUT, "norerank_%s.npz" % name), tgt=tgt, rho=rho, eps_bits=np.float16(eps).view(np.uint16),
                            count=cnt, top_num=top, labels=labels.astype(np.int64), sha_euclid=sha(e),
                            **({"euclid": e} if N <= 256 else {}))

    # ---- DBSCAN-only fixtures (sklearn is the oracle's oracle): non-zero diagonal, border
    # conflicts, all-noise, single cluster, duplicates
    rng = np.random.default_rng(7)
    mats, epss, labs = [], [], []
    for case in range(8):
        N = 120
        P = rng.random((N, 2)) * (1.0 if case % 2 == 0 else 3.0)
        D = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
        D = D + np.diag(rng.random(N) * 0.05)          # non-zero diagonal like final_dist
        eps = [0.08, 0.35, 0.2, 0.3, 0.001, 10.0, 0.1, 0.45][case]
        if case == 6:
            D[5] = D[6]; D[:, 5] = D[:, 6]              # duplicate rows
        lab = DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(D)
        ol = ora.dbscan(D, eps, 4)
        ok = ok and beq(lab, ol)
        print("dbscan case %d: oracle==sklearn %s clusters=%d noise=%d" % (case, beq(lab, ol), lab.max() + 1, int((lab < 0).sum())))
        mats.append(D); epss.append(eps); labs.append(lab.astype(np.int64))
    np.savez_compressed(os.path.join(OUT, "dbscan_cases.npz"), D=np.stack(mats), eps=np.array(epss), labels=np.stack(labs))

    ok = init_fixture(mod) and ok
    ok = embed_fixture() and ok
    ok = embed_fixture_wide() and ok
    ok = eval_fixture() and ok
    ok = pairwise_fixture() and ok
    ok = jpeg_fixture() and ok
    ok = plain_fixture() and ok
    ok = wide_fixture(mod) and ok
    print("ALL OK" if ok else "ORACLE MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
