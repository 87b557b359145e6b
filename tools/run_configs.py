#!/usr/bin/env python3
"""BASELINE.json configs[1..4] on ONE MI355X (the 8-GPU legs are the driver's): wall time per stage with a device
sync at the stage boundaries only.  Synthetic tracks as in bench.py / SURVEY.md 8d.  Prints one JSON line per config."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def group_roofline(src, tgt, lam, rho, splits=1):
    """one more (untimed by the wall clock) grouping leg with HIP events around every C-ABI launch -> the same `roofline_kernels` /
    `roofline_k5_k12` objects as bench.py (tools/roofline.py), at this configuration's size, plus the full per-entry-point table"""
    from ssg_amd import rerank, cluster, _lib
    from roofline import KernelTimer, grouping_roofline
    real = _lib.lib()
    timer = KernelTimer(real)
    _lib._lib = timer
    try:
        timer.on = True
        os.environ["SSG_RERANK_OVERLAP"] = "0"                  # one stream: an event pair brackets a launch that runs alone
        h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=lam, keep_euclid=False, validate=False)
        res = cluster.eps_rule_dbscan(h, rho, min_samples=4)    # the product's chain (generate_selflabel, iteration 0): one read-back
        tot = timer.totals()
    finally:
        _lib._lib = real
        os.environ.pop("SSG_RERANK_OVERLAP", None)
    split = cluster.sparse_row_split(h, rho, res[0])            # (untimed) which path the rows of the two sparse passes took
    del h
    N, Ns = tgt.shape[0], src.shape[0]
    kernels, k5 = grouping_roofline(tot, N, N, Ns, 1, 1, d=tgt.shape[1], row_split=split)
    table = {k: {"launches": n, "ms": round(ms, 3)} for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])}
    return {"roofline_kernels": kernels, "roofline_k5_k12": k5, "abi_launch_ms": table, "all_abi_ms": round(sum(ms for _, ms in tot.values()), 3)}


def group(src, tgt, lam, rho, no_rerank=False, reps=2):
    from ssg_amd import rerank, cluster
    best = None
    for _ in range(reps):
        t0 = sync()
        h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=lam, no_rerank=no_rerank, keep_euclid=no_rerank)
        t1 = sync()
        eps, cnt, top = cluster.eps_rule(h, rho)
        t2 = sync()
        lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
        t3 = sync()
        r = dict(dist_ms=round((t1 - t0) * 1e3, 2), eps_ms=round((t2 - t1) * 1e3, 2), dbscan_ms=round((t3 - t2) * 1e3, 2),
                 total_s=round(t3 - t0, 4), eps=float(eps), clusters=int(lab.max() + 1), noise=int((lab < 0).sum()))
        # the same leg the way the product runs it (no sync between the stages, eps rule + DBSCAN as one device chain)
        del h
        t4 = sync()
        h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=lam, no_rerank=no_rerank, keep_euclid=no_rerank, validate=False)
        e2, _, _, l2, _ = cluster.eps_rule_dbscan(h, rho, min_samples=4)
        t5 = sync()
        r["fused_total_s"] = round(t5 - t4, 4)
        r["fused_equal"] = bool(float(e2) == float(eps)) and bool((l2 == lab).all())
        del h
        if best is None or r["total_s"] < best["total_s"]:
            best = r
    return best


def embed_rate(n_images, num_split, batch=512):
    import ssg_amd
    dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=num_split, cluster=False, seed=1).cuda().eval()
    x = torch.randn(batch, 3, 256, 128, device=dev)
    m.embed_with_flip(x)
    t0 = sync()
    done = 0
    while done < n_images:
        b = min(batch, n_images - done)
        m.embed_with_flip(x[:b]); done += b
    t1 = sync()
    return round(t1 - t0, 3), round(n_images / (t1 - t0), 1)


def main():
    from conftest import clustered
    dev = torch.device("cuda", 0)
    which = sys.argv[1:] or ["0", "1", "2", "3", "4"]
    if "0" in which:
        # BASELINE configs[0]: N = 2 000 Track-I images (N(0,1) pixels) through the whole selftraining.py chain on the GPU: extract_features ->
        # no-rerank pairwise L2 -> eps rule -> DBSCAN (the reference runs this configuration on the CPU; tests/test_gpu_chain.py checks it against the oracle)
        import ssg_amd
        g = torch.Generator(device=dev).manual_seed(1)
        imgs = torch.randn(2000, 3, 256, 128, generator=g, device=dev)
        m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False).cuda().eval()
        ssg_amd.extract_embeddings(m, ssg_amd.TensorBatchLoader(imgs[:500], 500))
        t0 = sync(); feats, _, _ = ssg_amd.extract_embeddings(m, ssg_amd.TensorBatchLoader(imgs, 1000)); t1 = sync()
        print(json.dumps({"config": 0, "what": "N=2000 Track-I images: HIP embed -> pairwise L2 (half) -> eps -> DBSCAN, no re-rank", "embed_s": round(t1 - t0, 4),
                          "embed_img_s": round(2000 / (t1 - t0), 1), **group(feats[:500].contiguous(), feats, 0.1, 1.6e-3, no_rerank=True)}), flush=True)
        del imgs, feats, m
    if "1" in which or "2" in which:
        tgt = torch.from_numpy(clustered(16000, 2048, 1)).to(dev); src = torch.from_numpy(clustered(12936, 2048, 2, intra=0.7)).to(dev)
        es, rate = embed_rate(16000 + 12936, 1)
        if "1" in which:
            print(json.dumps({"config": 1, "what": "N=16000 embed + pairwise L2 (half) + eps + DBSCAN, no re-rank", "embed_s": es, "embed_img_s": rate,
                              **group(src, tgt, 0.3, 1.6e-3, no_rerank=True)}), flush=True)
        if "2" in which:
            for lam in (0.3, 0.1):
                print(json.dumps({"config": 2, "what": "N=16000 full k-reciprocal re-rank lambda=%.1f + eps + DBSCAN" % lam, "embed_s": es, "embed_img_s": rate,
                                  **group(src, tgt, lam, 1.6e-3)}), flush=True)
        del tgt, src
    if "3" in which:
        es, rate = embed_rate(30000 + 12936, 2)
        tot = 0.0; parts = []
        for s in range(3):
            tgt = torch.from_numpy(clustered(30000, 2048, 10 + s)).to(dev); src = torch.from_numpy(clustered(12936, 2048, 20 + s, intra=0.7)).to(dev)
            r = group(src, tgt, 0.3, 1.6e-3, reps=1 if s else 2); parts.append(r); tot += r["total_s"]
            if s == 0:
                roof = group_roofline(src, tgt, 0.3, 1.6e-3)
            del tgt, src
        print(json.dumps({"config": 3, "what": "N=30000 (Ns=12936), 3 feature splits, ONE GPU: embed (num_split=2) + 3 x (re-rank + eps + DBSCAN)",
                          "embed_s": es, "embed_img_s": rate, "grouping_s_3_splits": round(tot, 4), "iteration_s": round(es + tot, 3), "splits": parts,
                          "per_kernel_of_one_split": roof}), flush=True)
    if "4" in which:
        N = 128000
        tgt = torch.from_numpy(clustered(N, 2048, 31)).to(dev); src = torch.from_numpy(clustered(12936, 2048, 32, intra=0.7)).to(dev)
        r = group(src, tgt, 0.3, 1.6e-3, reps=2)       # best of two: the first call pays for 64+ GB of fresh allocations
        nn2 = 8.0 * N * N
        roof = group_roofline(src, tgt, 0.3, 1.6e-3)
        print(json.dumps({"config": 4, "what": "N=128000 re-rank + eps + DBSCAN on ONE GPU (32 GB half D + 32 GB half J'), second call", **r,
                          "k5_k12_algorithmic_GB": round(nn2 / 1e9, 1), "per_kernel": roof}), flush=True)


if __name__ == "__main__":
    main()
