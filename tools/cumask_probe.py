#!/usr/bin/env python3
"""Two embedding forwards on two CU-masked HIP streams (development probe): each stream owns half of the CUs, so an HBM-bound
launch of one forward runs beside an MFMA-bound launch of the other instead of after it."""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def masked_stream(hip, words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(st.value)


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
    hip = ctypes.CDLL("libamdhip64.so")

    def run(streams, n, stagger):
        t0 = time.time()
        if stagger:     # half a forward of head start for stream 0: its first forward alone, then alternate
            with torch.cuda.stream(streams[0]):
                m._fmap(x, flip=False)
        for i in range(n):
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    m._fmap(x, flip=bool(k))
        torch.cuda.synchronize()
        return (time.time() - t0) / (n + (0.5 if stagger else 0))

    def single(n):
        t0 = time.time()
        for i in range(n):
            m._fmap(x, flip=False); m._fmap(x, flip=True)
        torch.cuda.synchronize()
        return (time.time() - t0) / n

    full = [0xFFFFFFFF] * 8
    configs = {
        "two plain streams": None,
        "halves (CUs 0-127 | 128-255)": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
        "alternating CUs": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
        "alternating 32-CU words": ([0xFFFFFFFF, 0] * 4, [0, 0xFFFFFFFF] * 4),
        "both unmasked ext streams": (full, full),
    }
    dt = single(a.iters); dt = single(a.iters)
    print("%-36s %.2f ms per (orig + flip) of %d images -> %.0f img/s" % ("single stream", dt * 1e3, a.B, a.B / dt))
    for name, masks in configs.items():
        st = [torch.cuda.Stream(), torch.cuda.Stream()] if masks is None else [masked_stream(hip, list(mk)) for mk in masks]
        for s_ in st:
            with torch.cuda.stream(s_):
                m._fmap(x)
        torch.cuda.synchronize()
        for stagger in (False, True):
            run(st, 2, stagger)
            dt = run(st, a.iters, stagger)
            print("%-36s %s %.2f ms per (orig + flip) -> %.0f img/s" % (name, "staggered" if stagger else "in phase ", dt * 1e3, a.B / dt))
    # one forward alone on half of the CUs: how much of the whole-chip rate do the HBM-bound launches keep?
    st = masked_stream(hip, [0xFFFFFFFF] * 4 + [0] * 4)
    with torch.cuda.stream(st):
        m._fmap(x); torch.cuda.synchronize()
        t0 = time.time()
        for i in range(a.iters):
            m._fmap(x)
        torch.cuda.synchronize()
        half = (time.time() - t0) / a.iters
    whole = single(a.iters) / 2
    print("one forward on 128 CUs: %.2f ms (whole chip: %.2f ms)" % (half * 1e3, whole * 1e3))


if __name__ == "__main__":
    main()
