#!/usr/bin/env python3
"""shape of the Jaccard walk at the bench size (development aid): columns per V_qe row, inverted-list lengths (the row kernel takes 64
entries of a list per step; longer lists go through a dependent q_idx -> colptr -> inv_row load chain), touched columns per row"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 16000)); Ns = 12936 * N // 16000
track = os.environ.get("TRACK", "hard")
gen = synth.hard_clustered if track == "hard" else synth.clustered
src = torch.from_numpy(gen(Ns, 2048, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(gen(N, 2048, 1)).to(dev)
st = {}
h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, keep_euclid=False, stages=st)
qn = st["q_nnz"].long().cpu(); cp = st["colptr"].long().cpu(); ln = cp[1:N + 1] - cp[:N]
def q(t, ps=(0.5, 0.9, 0.99)):
    t = t.double()
    return "mean %.1f, median %d, p90 %d, p99 %d, max %d" % (t.mean(), *[int(torch.quantile(t, p)) for p in ps], int(t.max()))
print("track %s N=%d" % (track, N))
print("columns per row (q_nnz):", q(qn))
print("inverted-list length:", q(ln), "| lists > 64: %.1f %%, > 128: %.1f %%" % (100.0 * (ln > 64).double().mean(), 100.0 * (ln > 128).double().mean()))
# per row: how many of ITS columns have long lists (each costs the dependent chain)
qi = st["q_idx"].long().cpu(); capQ = qi.shape[1] if qi.dim() == 2 else qi.numel() // N
qi = qi.view(N, capQ)
mask = torch.arange(capQ)[None, :] < qn[:, None]
lens = ln[qi.clamp(0, N - 1)] * mask
print("per row: columns with a list > 64:", q((lens > 64).sum(1)), "; entries walked per row:", q(lens.sum(1)))
sp = getattr(h, "sparse", None)
if sp is not None:
    print("touched columns per row (sparse segment length):", q(sp["seg_len"].long().cpu()))
