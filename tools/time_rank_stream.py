#!/usr/bin/env python3
"""introsort ranking: rows in LDS vs the streamed path forced onto the same rows (development aid): usage time_rank_stream.py N ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import clustered  # noqa: E402
from ssg_amd import rerank, _lib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for N in [int(a) for a in sys.argv[1:]] or [16000, 24000, 30000]:
        tgt = rerank._as_dev_f32(clustered(N, 128, 1), dev)
        D, rowmax, _ = rerank._original_distance(_lib.lib(), tgt, 0, N, float(tgt.abs().max()), _lib.stream())
        res = {}
        for name, force in (("lds", False), ("stream", True)):
            r = rerank.initial_rank(D, rowmax, N, N, 21, "introsort", force_arena=force)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = rerank.initial_rank(D, rowmax, N, N, 21, "introsort", force_arena=force)
            e1.record(); torch.cuda.synchronize()
            res[name] = (e0.elapsed_time(e1) / 5, r)
        print("N=%d  rows in LDS %.3f ms   streamed (cap %s) %.3f ms   equal %s" % (N, res["lds"][0], os.environ.get("SSG_INTRO_STREAM_CAP", "natural"), res["stream"][0],
                                                                                 bool(torch.equal(res["lds"][1], res["stream"][1]))), flush=True)


if __name__ == "__main__":
    main()
