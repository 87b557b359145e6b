#!/usr/bin/env python3
"""Two forwards on two HIP streams: in phase (the product's embed_with_flip) against STAGGERED by half a network, so that the
HBM-bound half (stem, layer1, layer2) of one forward runs beside the matrix-bound half (layer3, layer4) of the other.
Lock step: stream A enters layer3 of forward i  <->  stream B enters layer1 of forward i, and the other way round."""
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
    ap.add_argument("--at", type=int, default=7, help="block index where the second half starts (7 = layer3)")
    a = ap.parse_args()
    import ssg_amd
    dev = torch.device("cuda", 0)
    m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False).cuda().eval()
    x = torch.randn(a.B, 3, 256, 128, device=dev)
    m.embed_with_flip(x); torch.cuda.synchronize()

    def in_phase():
        for _ in range(a.iters):
            m.embed_with_flip(x, check_overflow=False)

    sa, sb = m._side_streams()

    def serial():
        for _ in range(a.iters):
            m.pooled(*m._fmap(x, flip=False)); m.pooled(*m._fmap(x, flip=True))

    def free_running():
        # no coupling at all: each stream runs its forwards back to back, B starts when A reaches layer3 the first time
        ev = torch.cuda.Event()
        with torch.cuda.stream(sa):
            m.pooled(*m._fmap(x, flip=False, hooks={a.at: lambda: ev.record(sa)}))
        sb.wait_event(ev)
        for i in range(a.iters):
            with torch.cuda.stream(sb):
                m.pooled(*m._fmap(x, flip=True))
            if i + 1 < a.iters:
                with torch.cuda.stream(sa):
                    m.pooled(*m._fmap(x, flip=False))
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)

    def lock_step():
        # A: first half of i+1 may start once B entered its second half of i; B: first half of i starts once A entered its second half of i
        a_mid = [torch.cuda.Event() for _ in range(a.iters)]
        b_mid = [torch.cuda.Event() for _ in range(a.iters)]
        for i in range(a.iters):
            with torch.cuda.stream(sa):
                if i > 0:
                    sa.wait_event(b_mid[i - 1])
                ra = m.pooled(*m._fmap(x, flip=False, hooks={a.at: (lambda i=i: a_mid[i].record(sa))}))
            with torch.cuda.stream(sb):
                sb.wait_event(a_mid[i])
                rb = m.pooled(*m._fmap(x, flip=True, hooks={a.at: (lambda i=i: b_mid[i].record(sb))}))
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)

    for name, fn in (("serial", serial), ("in_phase", in_phase), ("free_running", free_running), ("lock_step", lock_step),
                     ("in_phase", in_phase), ("lock_step", lock_step), ("free_running", free_running)):
        fn(); torch.cuda.synchronize()
        t0 = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t0
        print("%-13s B=%d: %.2f ms per image pair batch -> %.0f img/s" % (name, a.B, dt / a.iters * 1e3, a.B * a.iters / dt), flush=True)


if __name__ == "__main__":
    main()
