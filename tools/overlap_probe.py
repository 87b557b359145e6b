#!/usr/bin/env python3
"""Do two embedding forwards on two HIP streams overlap usefully?  (development probe: the original and the flipped forward of
`embed_with_flip` are independent; HBM-bound launches of one could run under MFMA-bound launches of the other)"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=6)
    a = ap.parse_args()
    import ssg_amd
    dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False, pretrained=False).cuda().eval()
    x = torch.randn(a.B, 3, 256, 128, device=dev)
    m._fmap(x); m._fmap(x, flip=True); torch.cuda.synchronize()

    def run_single(n):
        t0 = time.time()
        for i in range(n):
            m._fmap(x, flip=False); m._fmap(x, flip=True)
        torch.cuda.synchronize()
        return (time.time() - t0) / n

    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in s:
        with torch.cuda.stream(st):
            m._fmap(x)
    torch.cuda.synchronize()

    def run_dual(n, offset):
        """two forwards in flight; with `offset` the second stream starts half a forward later (hook after layer2 = block 7)"""
        t0 = time.time()
        if offset:
            # crude half-forward offset: stream 1 first runs a forward alone up to its middle by being launched first, and every
            # later forward of stream k waits for the middle event of the other stream's previous forward
            mid = [None, None]
            orig = m._conv
            for i in range(n):
                for k in (0, 1):
                    with torch.cuda.stream(s[k]):
                        if mid[1 - k] is not None:
                            s[k].wait_event(mid[1 - k])
                        cnt = [0]

                        def hooked(L, xx, f, res=None, relu=True, out_split=False, ovf=None, _k=k, _cnt=cnt):
                            out = orig(L, xx, f, res, relu, out_split, ovf)
                            _cnt[0] += 1
                            if _cnt[0] == 6:          # after layer3's first launches
                                ev = torch.cuda.Event(); ev.record(s[_k]); mid[_k] = ev
                            return out
                        type(m)._conv = staticmethod(hooked)
                        m._fmap(x, flip=bool(k))
            type(m)._conv = staticmethod(orig)
        else:
            for i in range(n):
                for k in (0, 1):
                    with torch.cuda.stream(s[k]):
                        m._fmap(x, flip=bool(k))
        torch.cuda.synchronize()
        return (time.time() - t0) / n

    for name, f in (("single stream", lambda: run_single(a.iters)), ("two streams", lambda: run_dual(a.iters, False)),
                    ("two streams, half-forward offset", lambda: run_dual(a.iters, True)), ("single stream", lambda: run_single(a.iters))):
        f()
        dt = f()
        print("%-34s %.2f ms per (orig + flip) of %d images -> %.0f img/s" % (name, dt * 1e3, a.B, a.B / dt))


if __name__ == "__main__":
    main()
