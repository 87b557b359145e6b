#!/usr/bin/env python3
"""Dense passes of the eps rule / region query alone (development aid): ssg_eps_compact_below (the one full pass over the strict upper
triangle, N^2 bytes) and ssg_region_query_dev (2 N^2 bytes) on a re-ranked handle, HIP events over `reps` launches.
usage: time_compact.py [N ...]   (SSG_LAMBDA, default 0.1: Track-G at lambda = 0.1 puts the threshold above every row floor, i.e. all rows dense)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank, _lib
from ssg_amd._lib import check, ptr, stream
dev = torch.device("cuda", 0)
L = _lib.lib()
lam = float(os.environ.get("SSG_LAMBDA", "0.1"))
for N in [int(a) for a in sys.argv[1:]] or [30000]:
    tgt = torch.from_numpy(synth.clustered(N, 512, 1)).to(dev); src = torch.from_numpy(synth.clustered(4000, 512, 2, intra=0.7)).to(dev)
    h = rerank.re_ranking_device(src, tgt, lambda_value=lam, keep_euclid=False)
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    rho = 1.6e-3
    z = torch.zeros(2 * 4097 + 5 + 1, dtype=torch.int64, device=dev)
    check(L.ssg_eps_sample_threshold(*args, max(1, N // 192), 1.3 * rho, ptr(z[:8194]), ptr(z[8194:8199]), ptr(z[8199:]), None, stream()), "thr")
    thr3 = z[8194:8199]
    top = int(np.round(rho * (N * (N - 1) // 2)))
    n_cap = 1 << (2 * top + (1 << 16) - 1).bit_length()
    buf = torch.empty(n_cap, dtype=torch.int64, device=dev)
    reps = 10
    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    cur = torch.zeros(3, dtype=torch.int64, device=dev)
    def compact():
        cur.zero_()
        check(L.ssg_eps_compact_below(*args, ptr(thr3), ptr(buf), n_cap, ptr(cur), stream()), "compact")
    t_c = timed(compact)
    got = int(cur[0].item())
    if os.environ.get("SSG_TC_NOCAND", "0") == "1":       # the streaming rate of the pass alone: a threshold no element lies below
        keep = thr3.clone(); thr3[0] = int(np.float32(1e-30).view(np.uint32))
        t_0 = timed(compact); thr3.copy_(keep)
        print("N=%d: the same pass with no candidate: %.3f ms = %.2f TB/s" % (N, t_0, N * N / t_0 / 1e9), flush=True)
    eps = torch.tensor([float(np.uint32(int(thr3[0].item()) & 0xffffffff).view(np.float32)) * 0.97, 0.0], dtype=torch.float64, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev); ecap = 256 * N
    edges = torch.empty((ecap, 2), dtype=torch.int32, device=dev); ecur = torch.zeros(2, dtype=torch.int64, device=dev)
    def region():
        ecur.zero_()
        check(L.ssg_region_query_dev(ptr(h.M), ptr(h.v), N, 0, N, 0, h.lambda_value, ptr(eps), ptr(cnt), ptr(edges), ecap, ptr(ecur), stream()), "rq")
    t_r = timed(region)
    print("N=%d lambda=%.1f: dense eps_compact_below %.3f ms = %.2f TB/s of N^2 bytes (%d keys); dense region_query %.3f ms = %.2f TB/s of 2 N^2 bytes (%d hits)"
          % (N, lam, t_c, N * N / t_c / 1e9, got, t_r, 2.0 * N * N / t_r / 1e9, int(ecur[0].item())), flush=True)
    del h, buf, edges
