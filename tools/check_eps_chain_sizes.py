#!/usr/bin/env python3
"""fused eps rule + DBSCAN chain against the two-call path at sizes around the sample sort's window (development aid / GPU check):
N = 3000 .. 19000 use the sample sort, 21000 the bitonic network; both rho of the reference's configs."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank, cluster
dev = torch.device("cuda", 0)
bad = 0
for N in (3000, 6000, 12000, 17000, 19000, 21000):
    Ns = 12936 * N // 16000
    src = torch.from_numpy(synth.hard_clustered(Ns, 2048, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(synth.hard_clustered(N, 2048, 1)).to(dev)
    for lam in (0.3, 0.0):
        h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=lam, keep_euclid=False, validate=False)
        for rho in (1.6e-3, 2.0e-3):
            os.environ["SSG_EPS_FUSED"] = "1"
            a = cluster.eps_rule_dbscan(h, rho, min_samples=4)
            os.environ["SSG_EPS_FUSED"] = "0"
            b = cluster.eps_rule_dbscan(h, rho, min_samples=4)
            ok = a[0] == b[0] and a[1:3] == b[1:3] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
            bad += not ok
            print("N=%d lambda=%.1f rho=%.1e: eps %.9f top %d clusters %d -> %s" % (N, lam, rho, a[0], a[2], int(a[3].max()) + 1, "equal" if ok else "DIFFERENT"))
    del h, src, tgt
print("mismatches:", bad)
sys.exit(1 if bad else 0)
