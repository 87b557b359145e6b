#!/usr/bin/env python3
"""embedding throughput by batch size, and bit equality of the features of the first images with the B = 256 result (development aid)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssg_amd
dev = torch.device("cuda", 0)
m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False).cuda().eval()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1024, 3, 256, 128, generator=g, device=dev)
ref = m.embed_with_flip(x[:256]).clone()
for B in [int(a) for a in sys.argv[1:]] or [1000, 1008, 1016, 1023, 1024]:
    try:
        f = m.embed_with_flip(x[:B]); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(4):
            f = m.embed_with_flip(x[:B])
        torch.cuda.synchronize(); dt = (time.time() - t0) / 4
        print("B=%d: %.2f ms per call, %.0f img/s, first 256 images bit-equal to the B=256 call: %s, last-image finite: %s" % (
            B, dt * 1e3, B / dt, bool(torch.equal(f[:256], ref)), bool(torch.isfinite(f[-1]).all())), flush=True)
    except Exception as e:       # noqa: BLE001
        print("B=%d failed: %s" % (B, str(e)[:200]), flush=True)
