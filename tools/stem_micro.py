#!/usr/bin/env python3
"""Fused stem (ssg_stem_pool_nchw_x) vs the three-launch path: equality + timing (development aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
from ssg_amd import _lib
from ssg_amd._lib import check, ptr, stream


def main():
    B = int(os.environ.get("MICRO_B", "512"))
    L = _lib.lib(); dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False).cuda().eval()
    net = m._prepare(); st = net["stem"]
    x = torch.randn(B, 3, 256, 128, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)

    def unfused(flip):
        x4 = torch.empty((B, 256, 128, 4), dtype=torch.float32, device=dev)
        check(L.ssg_nchw_to_nhwc4_h4l4(ptr(x), ptr(x4), B, 256, 128, flip, stream()), "nhwc4")
        y = m._conv(L, x4, st, out_split=True, ovf=ovf)
        p = torch.empty((B, 64, 32, 64), dtype=torch.float32, device=dev)
        check(L.ssg_maxpool3x3s2_h8l8(ptr(y), ptr(p), B, 128, 64, 64, stream()), "pool")
        return p

    def fused(flip):
        y = torch.empty((B, 64, 32, 64), dtype=torch.float32, device=dev)
        check(L.ssg_stem_pool_nchw_x(ptr(x), flip, ptr(st.w), ptr(st.bias), ptr(st.cscale), ptr(y), B, 256, 128, ptr(ovf), stream()), "stem_pool")
        return y

    for flip in (0, 1):
        a = unfused(flip); b = fused(flip); torch.cuda.synchronize()
        neq = (a.view(torch.int32) != b.view(torch.int32))
        print("flip=%d: words differing %d of %d" % (flip, int(neq.sum()), neq.numel()))
        if int(neq.sum()):
            bad = neq.view(B, 64, 32, 64).any(dim=3)
            print("  bad pixels per pooled row (image 0):", bad[0].sum(dim=1).tolist())
            print("  bad per pooled col (image 0):", bad[0].sum(dim=0).tolist())
    for name, f in (("unfused", unfused), ("fused", fused)):
        f(0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f(0)
        e1.record(); e1.synchronize()
        print("  %-8s %.3f ms" % (name, e0.elapsed_time(e1) / 10))


if __name__ == "__main__":
    main()
