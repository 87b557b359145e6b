#!/usr/bin/env python3
"""Fused identity bottleneck (ssg_bottleneck_nhwc_x) vs the three-launch path: equality + timing (development aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
from ssg_amd import _lib, resnet
from ssg_amd._lib import check, ptr, stream


def main():
    B = int(os.environ.get("MICRO_B", "512")); blk_i = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    L = _lib.lib(); dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False).cuda().eval()
    net = m._prepare(); blk = net["blocks"][blk_i]
    C = blk["c1"].cin; CO = blk["c3"].cout
    H, W = {256: 64, 512: 32, 1024: 16, 2048: 8}[CO], {256: 32, 512: 16, 1024: 8, 2048: 4}[CO]
    x = torch.relu(torch.randn(B, H, W, C, device=dev)) * 0.7
    xs = torch.empty_like(x); check(L.ssg_h8l8_encode(ptr(x), ptr(xs), x.numel(), 1.0, stream()), "enc")
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)

    def unfused():
        o = m._conv(L, xs, blk["c1"], out_split=True, ovf=ovf)
        o = m._conv(L, o, blk["c2"], out_split=True, ovf=ovf)
        if blk["ds"] is not None:
            return m._conv_dual(L, o, xs, blk["c3"], blk["ds"], out_split=True, ovf=ovf)
        return m._conv(L, o, blk["c3"], res=xs, relu=True, out_split=True, ovf=ovf)

    def fused():
        return m._bottleneck(L, xs, blk, ovf)

    a = unfused(); b = fused(); torch.cuda.synchronize()
    assert b is not None, "no fused kernel for this block"
    ai, bi = a.view(torch.int32), b.view(torch.int32)
    neq = int((ai != bi).sum())
    da = torch.empty_like(a); db = torch.empty_like(b)
    check(L.ssg_h8l8_decode(ptr(a), ptr(da), a.numel(), 1.0, stream()), "dec"); check(L.ssg_h8l8_decode(ptr(b), ptr(db), b.numel(), 1.0, stream()), "dec")
    print("block %d  B=%d %dx%dx%d: words differing %d of %d; max |diff| %.3e (max |ref| %.3e); ovf %d" % (blk_i, B, H, W, C, neq, ai.numel(), float((da - db).abs().max()), float(da.abs().max()), int(ovf.item())))
    if neq:
        idx = (ai != bi).nonzero()[:5]
        print("first mismatches (b,y,x,c):", idx.tolist())
        bad = (ai != bi).view(B, H, W, CO).any(dim=3)
        print("bad pixels per image row (image 0):", bad[0].sum(dim=1).tolist())
    for name, f in (("unfused", unfused), ("fused", fused)):
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("  %-8s %.3f ms  (%.2f TB/s on read x + write out)" % (name, ms, 4.0 * (x.numel() + a.numel()) / ms / 1e9))


if __name__ == "__main__":
    main()
