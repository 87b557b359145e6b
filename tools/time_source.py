#!/usr/bin/env python3
"""source term (ssg_source_rowmin_filtered1 through rerank.source_vector) at the bench shape, HIP events over 20 calls (development aid)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank
dev = torch.device("cuda", 0)
src = torch.from_numpy(synth.hard_clustered(12936, 2048, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(synth.hard_clustered(16000, 2048, 1)).to(dev)
stats = rerank.range_stats(tgt, src)
ref = rerank.source_vector(src, tgt, stats=stats).clone()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    r = rerank.source_vector(src, tgt, stats=stats)
e1.record(); torch.cuda.synchronize()
print("source term: %.3f ms per call (SSG_SB_FUSED_ENC=%s), result sum %d" % (e0.elapsed_time(e1) / 20, os.environ.get("SSG_SB_FUSED_ENC", "1"), int(r.long().sum())))
