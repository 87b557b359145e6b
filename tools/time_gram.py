#!/usr/bin/env python3
"""int8 Gram (ssg_gram_i8_encode + ssg_sqdist_self_i8 through rerank._original_distance), HIP events over 20 calls (development aid):
usage time_gram.py [N ...]; SSG_I8_DMA=0 selects the register-staged kernel"""
import os, sys, hashlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank, _lib
dev = torch.device("cuda", 0)
for N in [int(a) for a in sys.argv[1:]] or [16000]:
    tgt = torch.from_numpy(synth.hard_clustered(N, 2048, 1)).to(dev)
    mx = float(tgt.abs().max())
    D, rowmax, _ = rerank._original_distance(_lib.lib(), tgt, 0, N, mx, _lib.stream())
    torch.cuda.synchronize()
    reps = 20 if N <= 40000 else 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        D, rowmax, _ = rerank._original_distance(_lib.lib(), tgt, 0, N, mx, _lib.stream())
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha256(D[:4096].cpu().numpy().tobytes()).hexdigest()[:12] + ":" + hashlib.sha256(rowmax.cpu().numpy().tobytes()).hexdigest()[:12]
    print("N=%d int8 Gram (encode + multiply): %.3f ms per call (SSG_I8_DMA=%s)  sha %s" % (N, e0.elapsed_time(e1) / reps, os.environ.get("SSG_I8_DMA", "1"), h), flush=True)
    del D, rowmax, tgt
