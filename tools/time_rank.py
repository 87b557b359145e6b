#!/usr/bin/env python3
"""Time the two K5 ranking kernels (introsort tie order = reference default; stable = opt-in) on Track-G data."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import clustered  # noqa: E402
from ssg_amd import rerank, _lib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    sizes = [int(a) for a in sys.argv[1:]] or [2000, 16000, 16522, 30000]
    for N in sizes:
        d = 128
        tgt = rerank._as_dev_f32(clustered(N, d, 1), dev)
        D, rowmax, flag = rerank._original_distance(_lib.lib(), tgt, 0, N, float(tgt.abs().max()), _lib.stream())
        K = 21
        out = {}
        for mode in ("stable", "introsort"):
            r = rerank.initial_rank(D, rowmax, N, N, K, mode)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                r = rerank.initial_rank(D, rowmax, N, N, K, mode)
            ev1.record(); torch.cuda.synchronize()
            out[mode] = (ev0.elapsed_time(ev1) / 5, r)
        diff = float((out["stable"][1] != out["introsort"][1]).any(dim=1).float().mean())
        print("N=%d K=%d  stable %.3f ms  introsort %.3f ms  (%.2f GB/s of the 2N^2 read)  rows whose top-K differs: %.1f%%" % (
            N, K, out["stable"][0], out["introsort"][0], 2.0 * N * N / out["introsort"][0] / 1e6, 100 * diff), flush=True)


if __name__ == "__main__":
    main()
