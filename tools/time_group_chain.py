#!/usr/bin/env python3
"""The grouping leg exactly as the bench's timed region runs it -- rerank.re_ranking_device (validate=False) followed by
cluster.eps_rule_dbscan (the device chain, ONE read-back) -- at the bench shape, `--reps` times (development aid; the command the round-6 PMC
passes profile: counters of the kernels that are actually timed).  --N 128000 --d 256 for the streamed introsort."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank, cluster
ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=16000); ap.add_argument("--Ns", type=int, default=12936); ap.add_argument("--d", type=int, default=2048)
ap.add_argument("--lam", type=float, default=0.3); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--track", default="hard")
a = ap.parse_args()
dev = torch.device("cuda", 0)
gen = synth.hard_clustered if a.track == "hard" else synth.clustered
src = torch.from_numpy(gen(a.Ns, a.d, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(gen(a.N, a.d, 1)).to(dev)
for r in range(a.reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=a.lam, keep_euclid=False, validate=False)
    eps, cnt, top, lab, core = cluster.eps_rule_dbscan(h, 1.6e-3, min_samples=4)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("rep %d: N=%d re-rank + eps rule + DBSCAN chain %.3f ms wall; eps %.6f, %d clusters, %d noise" % (r, a.N, (t1 - t0) * 1e3, eps, int(lab.max()) + 1, int((lab < 0).sum())), flush=True)
    del h
