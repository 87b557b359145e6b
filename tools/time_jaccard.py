#!/usr/bin/env python3
"""ssg_jaccard_rows2 at the bench shape, HIP events around the launch inside re_ranking_device (development aid; SSG_LIB_PATH selects an
A/B build of the same ABI)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import _lib, rerank
import roofline
dev = torch.device("cuda", 0)
timer = roofline.KernelTimer(_lib.lib()); _lib._lib = timer
N = int(os.environ.get("N", 16000)); Ns = 12936 * N // 16000
src = torch.from_numpy(synth.hard_clustered(Ns, 2048, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(synth.hard_clustered(N, 2048, 1)).to(dev)
rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, keep_euclid=False, validate=False)
torch.cuda.synchronize()
timer.on = True
for _ in range(10):
    rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, keep_euclid=False, validate=False)
tot = timer.totals()
n, ms = tot["ssg_jaccard_rows2"]
print("ssg_jaccard_rows2 (%s): %.4f ms per launch over %d launches" % (os.path.basename(os.environ.get("SSG_LIB_PATH", "product build")), ms / n, n))
