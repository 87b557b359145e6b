#!/usr/bin/env python3
"""Embedding throughput + per-layer conv timing on one GPU (development aid)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--layers", action="store_true")
    a = ap.parse_args()
    import ssg_amd
    from ssg_amd import _lib, resnet
    dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False).cuda().eval()
    x = torch.randn(a.B, 3, 256, 128, device=dev)
    m.embed_with_flip(x); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        f = m.embed_with_flip(x)
    torch.cuda.synchronize(); dt = (time.time() - t0) / a.iters
    print("B=%d embed_with_flip: %.1f ms -> %.0f img/s, %.1f TFLOP/s (10.68 GFLOP/img)" % (a.B, dt * 1e3, a.B / dt, a.B * 10.68e9 / dt / 1e12))
    if a.layers:
        L = _lib.lib()
        rows = []
        orig = resnet.ResNet._conv

        def timed(L_, x_, f, res=None, relu=True, out_split=False, ovf=None):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = orig(L_, x_, f, res, relu, out_split, ovf); e1.record(); e1.synchronize()
            B, H, W, _ = x_.shape
            flop = 2.0 * out.numel() * f.k * f.k * (3 if f.cin == 4 else f.cin)
            byt = 4.0 * (x_.numel() + out.numel() * (2 if res is not None else 1) + f.w.numel())
            rows.append((H, W, f.cin, f.cout, f.k, f.stride, e0.elapsed_time(e1), flop, byt))
            return out
        orig_d = resnet.ResNet._conv_dual

        def timed_d(L_, o, x_, c3, ds, out_split=False, ovf=None):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = orig_d(L_, o, x_, c3, ds, out_split, ovf); e1.record(); e1.synchronize()
            B, H, W, _ = o.shape
            flop = 2.0 * out.numel() * (c3.cin + ds.cin)
            byt = 4.0 * (o.numel() + x_.numel() / (ds.stride ** 2) + out.numel() + ds.w.numel())
            rows.append((H, W, c3.cin + ds.cin, c3.cout, 1, -ds.stride, e0.elapsed_time(e1), flop, byt))
            return out
        resnet.ResNet._conv = staticmethod(timed)
        resnet.ResNet._conv_dual = staticmethod(timed_d)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); m._fmap(x); e1.record(); e1.synchronize()
        print("  one forward (with per-layer syncs): %.2f ms" % e0.elapsed_time(e1))
        tot = sum(r[6] for r in rows)
        for r in rows:
            print("  %3dx%-3d cin=%4d cout=%4d k=%d s=%d  %7.3f ms  %6.1f TF/s  %6.2f TB/s" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7] / r[6] / 1e9, r[8] / r[6] / 1e9))
        print("  conv total %.1f ms for one forward of B=%d: %.1f TF/s" % (tot, a.B, sum(r[7] for r in rows) / tot / 1e9))


if __name__ == "__main__":
    main()
