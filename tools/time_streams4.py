#!/usr/bin/env python3
"""How many HIP streams for the forwards of one batch?  2 streams x 1000 images (the product's embed_with_flip) against 4 streams x 500
(original / flipped x two half batches) and 1 stream; same images, same launches per image (development probe)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
dev = torch.device("cuda", 0)
m = ssg_amd.create("resnet50", num_classes=0, num_split=2, cluster=False).cuda().eval()
B, iters = 1000, 8
x = torch.randn(B, 3, 256, 128, device=dev)
m.embed_with_flip(x); torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def run(parts):
    cur = torch.cuda.current_stream(dev)
    for _ in range(iters):
        k = 0
        for lo in range(0, B, B // parts):
            for flip in (False, True):
                st = streams[k % len(streams)] if parts * 2 > 1 else cur
                k += 1
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    m.pooled(*m._fmap(x[lo:lo + B // parts], flip=flip))
        for st in streams:
            cur.wait_stream(st)


for name, parts in (("2 streams x 1000", 1), ("4 streams x 500", 2), ("4 streams x 250 (8 forwards)", 4), ("2 streams x 1000", 1), ("4 streams x 500", 2)):
    run(parts); torch.cuda.synchronize()
    t0 = time.time(); run(parts); torch.cuda.synchronize(); dt = time.time() - t0
    print("%-30s %.2f ms per 1000 image pairs -> %.0f img/s" % (name, dt / iters * 1e3, B * iters / dt), flush=True)
