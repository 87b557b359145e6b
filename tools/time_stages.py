#!/usr/bin/env python3
"""Per-stage timing of the grouping path on one GPU (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=16000)
    ap.add_argument("--Ns", type=int, default=12936)
    ap.add_argument("--d", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--track", choices=("separable", "hard"), default="separable")
    ap.add_argument("--lam", type=float, default=0.1)
    a = ap.parse_args()
    from conftest import clustered, hard_clustered
    if a.track == "hard":
        clustered = hard_clustered      # noqa: F811  (bench.py's default track)
    from ssg_amd import rerank, cluster, _lib
    dev = torch.device("cuda", 0)
    tgt = torch.from_numpy(clustered(a.N, a.d, 1)).to(dev)
    src = torch.from_numpy(clustered(a.Ns, a.d, 2, intra=0.7)).to(dev)
    L = _lib.lib()
    orig = {}
    times = {}

    def wrap(name):
        fn = getattr(L, name)
        orig[name] = fn

        def timed(*args):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*args); e1.record(); e1.synchronize()
            times.setdefault(name, []).append(e0.elapsed_time(e1))
            return rc
        return timed

    class Proxy:
        def __getattr__(self, k):
            if k.startswith("ssg_") and k not in ("ssg_last_error", "ssg_krecip_row_capacity", "ssg_double_to_half_bits",
                                                  "ssg_eps_mean_workspace_bytes", "ssg_dbscan_cc_workspace_bytes", "ssg_version"):
                return wrap(k)
            return getattr(L, k)
    _lib._lib = Proxy()
    for rep in range(a.reps):
        times.clear()
        torch.cuda.synchronize(); t0 = time.time()
        h = rerank.re_ranking_device(src, tgt, lambda_value=a.lam, validate=False)
        torch.cuda.synchronize(); t1 = time.time()
        eps, cnt, top = cluster.eps_rule(h, 1.6e-3)
        torch.cuda.synchronize(); t2 = time.time()
        lab = cluster.DBSCAN(eps=eps, min_samples=4, metric="precomputed").fit_predict(h)
        torch.cuda.synchronize(); t3 = time.time()
        print("rep %d: rerank %.1f ms | eps %.1f ms (eps=%.5f top=%d) | dbscan %.1f ms (clusters=%d noise=%d)" % (
            rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, eps, top, (t3 - t2) * 1e3, lab.max() + 1, int((lab < 0).sum())))
        for k, v in times.items():
            print("    %-28s calls=%2d total=%9.3f ms" % (k, len(v), sum(v)))
    nn = a.N * a.N
    print("N^2*2B = %.1f MB; fp64 flop self=%.2f T cross=%.2f T" % (nn * 2 / 1e6, 2 * nn * a.d / 1e12, 2 * a.N * a.Ns * a.d / 1e12))


if __name__ == "__main__":
    main()
