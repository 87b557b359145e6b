#!/usr/bin/env python3
"""The three device-sized sorts of the eps rule on n keys inside a 2^k capacity (development aid): bitonic network, sample sort (LB = 10)
and the big sample sort (LB = 12).  usage: time_sort.py n [n ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssg_amd import _lib
from ssg_amd._lib import check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
for n in [int(a) for a in sys.argv[1:]] or [270000, 940000, 17000000]:
    n_cap = max(2048, 1 << (int(1.6 * n) - 1).bit_length())
    # keys like the eps rule's: float64 bit patterns of few-valued half sums crowded towards the threshold
    keys = ((torch.randint(0, 2048, (n_cap,), generator=g).double() / 2048.0) * 0.5 + 0.4 + torch.randint(0, 64, (n_cap,), generator=g).double() * 1e-3).view(torch.int64).to(dev)
    nd = torch.tensor([n, 0, 0], dtype=torch.int64, device=dev)
    ref = torch.sort(keys[:n]).values
    def timed(fn, reps=5):
        buf = keys.clone(); fn(buf); torch.cuda.synchronize()
        ok = bool(torch.equal(buf[:n], ref))
        ts = []
        for _ in range(reps):
            buf = keys.clone(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(buf); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts), ok
    fail = torch.zeros(1, dtype=torch.int64, device=dev)
    out = {"bitonic": timed(lambda b: check(L.ssg_sort_u64_dev(ptr(b), n_cap, ptr(nd), stream()), "s"))}
    w10 = torch.empty(int(L.ssg_samplesort_u64_workspace_bytes(n_cap)), dtype=torch.uint8, device=dev)
    if n <= 3000000:
        out["sample10"] = timed(lambda b: check(L.ssg_samplesort_u64_dev(ptr(b), n_cap, ptr(nd), ptr(w10), w10.numel(), ptr(fail), stream()), "s"))
    w12 = torch.empty(int(L.ssg_samplesort_u64_big_workspace_bytes(n_cap)), dtype=torch.uint8, device=dev)
    out["sample12"] = timed(lambda b: check(L.ssg_samplesort_u64_big_dev(ptr(b), n_cap, ptr(nd), ptr(w12), w12.numel(), ptr(fail), stream()), "s"))
    print("n=%d (capacity %d): " % (n, n_cap) + ", ".join("%s %.3f ms%s" % (k, v[0], "" if v[1] else " WRONG") for k, v in out.items()) + "; fail=%d" % int(fail.item()), flush=True)
