#!/usr/bin/env python3
"""Per-launch table of ONE embedding forward (development aid / profiles/): every C-ABI launch of `ResNet._fmap` in order, with
its HIP-event time (average over --reps forwards, events on the launch stream), the convolution flops it carries, its algorithmic
HBM bytes (inputs + outputs + residual + weights once, 4 B per value) and the two roofline fractions that follow (fp16 MFMA
peak / 3 products; 8 TB/s)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_TF, PEAK_GBS = 2516.6 / 3.0, 8000.0


class Rec:
    def __init__(self, L):
        self.L, self.rows, self.on = L, [], True

    def __getattr__(self, k):
        fn = getattr(self.L, k)
        if not self.on or not k.startswith("ssg_") or k.endswith("_supported") or k.endswith("_bytes") or k == "ssg_last_error":
            return fn

        def timed(*a):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*a); e1.record()
            self.rows.append((k, a, e0, e1))
            return rc
        return timed


def work(name, a, B):
    """(flop, bytes, label) of one launch from its ABI arguments (include/ssg_hip.h)"""
    v = lambda x: x.value if hasattr(x, "value") else x
    if name == "ssg_conv2d_nhwc_x":
        _, _, _, res, _, B_, H, W, Cin, Cout, KH, KW, stride, pad = [v(x) for x in a[:14]]
        OH = (H + 2 * pad - KH) // stride + 1; OW = (W + 2 * pad - KW) // stride + 1
        flop = 2.0 * B_ * OH * OW * Cout * KH * KW * Cin
        byt = 4.0 * (B_ * H * W * Cin + B_ * OH * OW * Cout * (2 if res else 1) + Cout * KH * KW * Cin)
        return flop, byt, "conv %dx%d %d->%d k%d s%d%s" % (H, W, Cin, Cout, KH, stride, " +res" if res else "")
    if name == "ssg_conv1x1_dual_nhwc_x":
        _, _, _, _, _, B_, H, W, C1, H2, W2, C2, stride2, Cout = [v(x) for x in a[:14]]
        flop = 2.0 * B_ * H * W * Cout * (C1 + C2)
        byt = 4.0 * (B_ * H * W * C1 + B_ * H * W * C2 + B_ * H * W * Cout + Cout * (C1 + C2))
        return flop, byt, "dual conv3|ds %dx%d (%d|%d)->%d s%d" % (H, W, C1, C2, Cout, stride2)
    if name == "ssg_stem_pool_nchw_x":
        B_, H, W = [v(x) for x in a[6:9]]
        flop = 2.0 * B_ * (H // 2) * (W // 2) * 64 * 147
        byt = 4.0 * (B_ * 3 * H * W + B_ * (H // 4) * (W // 4) * 64)
        return flop, byt, "stem+pool %dx%d" % (H, W)
    if name == "ssg_bottleneck_nhwc_x":
        B_, H, W, C, MID = [v(x) for x in a[11:16]]
        flop = 2.0 * B_ * H * W * (C * MID + 9 * MID * MID + MID * C)
        byt = 4.0 * (2 * B_ * H * W * C + C * MID + 9 * MID * MID + MID * C)
        return flop, byt, "bottleneck %dx%d C=%d mid=%d" % (H, W, C, MID)
    if name == "ssg_bottleneck_ds_nhwc_x":
        B_, H, W, CIN, C, MID = [v(x) for x in a[11:17]]
        flop = 2.0 * B_ * H * W * (CIN * MID + 9 * MID * MID + (MID + CIN) * C)
        byt = 4.0 * (B_ * H * W * (CIN + C) + CIN * MID + 9 * MID * MID + (MID + CIN) * C)
        return flop, byt, "bottleneck+ds %dx%d %d->%d mid=%d" % (H, W, CIN, C, MID)
    if name == "ssg_conv_pair_nhwc_x":
        M, K1, C, N2 = [v(x) for x in a[10:14]]
        flop = 2.0 * M * C * (K1 + N2)
        byt = 4.0 * (M * (K1 + 2 * C + N2) + C * (K1 + N2))
        return flop, byt, "pair conv3+res | next conv1 M=%d %d->%d->%d" % (M, K1, C, N2)
    return 0.0, 0.0, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import ssg_amd
    from ssg_amd import _lib
    dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False).cuda().eval()
    x = torch.randn(a.B, 3, 256, 128, device=dev)
    m._fmap(x); torch.cuda.synchronize()
    rec = Rec(_lib.lib()); _lib._lib = rec
    per = None
    for _ in range(a.reps):
        rec.rows = []
        m._fmap(x); torch.cuda.synchronize()
        ms = [r[2].elapsed_time(r[3]) for r in rec.rows]
        per = ms if per is None else [p + q for p, q in zip(per, ms)]
    rows = rec.rows
    _lib._lib = rec.L
    tot_ms = tot_fl = tot_by = 0.0
    out = []
    print("| # | launch | ms | TFLOP/s | frac MFMA(fp16/3) | alg GB | TB/s | frac HBM | bound by |")
    print("|---:|---|---:|---:|---:|---:|---:|---:|---|")
    for i, (r, p) in enumerate(zip(rows, per)):
        ms = p / a.reps
        fl, by, label = work(r[0], r[1], a.B)
        if fl == 0:
            continue
        tf, gbs = fl / ms / 1e9, by / ms / 1e6
        t_m, t_h = fl / (PEAK_TF * 1e9), by / (PEAK_GBS * 1e6)
        print("| %d | %s | %.3f | %.1f | %.3f | %.2f | %.2f | %.3f | %s (ideal %.3f ms) |" % (i, label, ms, tf, tf / PEAK_TF, by / 1e9, gbs / 1e3, gbs / PEAK_GBS,
                                                                                     "mfma" if t_m > t_h else "hbm", max(t_m, t_h)))
        out.append(dict(i=i, abi=r[0], label=label, ms=ms, flop=fl, bytes=by))
        tot_ms += ms; tot_fl += fl; tot_by += by
    print("\ntotal %.3f ms per forward of %d images: %.1f TFLOP/s (%.3f of fp16/3), %.2f TB/s algorithmic (%.3f of 8 TB/s); sum of per-launch roofline floors %.3f ms"
          % (tot_ms, a.B, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / PEAK_TF, tot_by / tot_ms / 1e9, tot_by / tot_ms / 1e6 / PEAK_GBS,
             sum(max(o["flop"] / (PEAK_TF * 1e9), o["bytes"] / (PEAK_GBS * 1e6)) for o in out)))
    if a.json:
        json.dump(out, open(a.json, "w"))


if __name__ == "__main__":
    main()
