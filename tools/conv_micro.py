#!/usr/bin/env python3
"""Micro-benchmark of single conv configurations (development aid)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
from ssg_amd import _lib
from ssg_amd._lib import check, ptr, stream

SPLIT = "split" in sys.argv


def run(B, H, W, Cin, Cout, k, stride, pad, res, reps=20):
    L = _lib.lib(); dev = torch.device("cuda", 0)
    x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, k * k * Cin, device=dev) * 0.02; b = torch.randn(Cout, device=dev)
    OH = (H + 2 * pad - k) // stride + 1; OW = (W + 2 * pad - k) // stride + 1
    out = torch.empty(B, OH, OW, Cout, device=dev); r = torch.randn_like(out) if res else None
    flags, sc = 0, 1.0
    if SPLIT:
        from ssg_amd.resnet import _h8l8
        xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc"); x = xs
        w = _h8l8(w.cpu() * 256.0).to(dev); sc = 1.0 / 256.0; flags = 3
        if res:
            rs = torch.empty_like(r); check(L.ssg_h8l8_encode(ptr(r), ptr(rs), r.numel(), 1.0, stream()), "enc"); r = rs
    f = lambda: check(L.ssg_conv2d_nhwc_x(ptr(x), ptr(w), ptr(b), ptr(r), ptr(out), B, H, W, Cin, Cout, k, k, stride, pad, 1, flags, sc, None, None, stream()), "conv")
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * OH * OW * Cout * k * k * Cin
    byt = 4.0 * (x.numel() + out.numel() * (2 if res else 1))
    print("B=%d %dx%d cin=%d cout=%d k=%d s=%d res=%d: %.3f ms %.1f TF/s %.2f TB/s" % (B, H, W, Cin, Cout, k, stride, int(res), ms, fl / ms / 1e9, byt / ms / 1e9))

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    B = int(os.environ.get("MICRO_B", "256"))
    if cfg == "c3": run(B, 16, 8, 256, 256, 3, 1, 1, False, 40)
    elif cfg == "c1": run(B, 16, 8, 1024, 256, 1, 1, 0, False, 40)
    elif cfg == "e1": run(B, 64, 32, 64, 256, 1, 1, 0, True, 40)
    elif cfg == "c3b": run(B, 32, 16, 128, 128, 3, 1, 1, False, 40)
    elif cfg == "c3a": run(B, 64, 32, 64, 64, 3, 1, 1, False, 40)
    elif cfg == "l4": run(B, 8, 4, 512, 512, 3, 1, 1, False, 40)
    elif cfg == "l4a": run(B, 8, 4, 2048, 512, 1, 1, 0, False, 40)
    elif cfg == "l4c": run(B, 8, 4, 512, 2048, 1, 1, 0, True, 40)
    elif cfg == "l3a": run(B, 16, 8, 1024, 256, 1, 1, 0, False, 40)
    elif cfg == "l3c": run(B, 16, 8, 256, 1024, 1, 1, 0, True, 40)
