#!/usr/bin/env python3
"""Throughput of the input side: `GpuBatchLoader(decode='gpu')` on Market-1501-sized JPEG files
(128 x 64, baseline 4:2:0, no restart markers -- what the dataset holds) and on the same pictures written with restart markers,
next to the reference's way (Pillow decode on the host, `decode='pillow'`), with the stages timed on their own:
  read      open + read of every file (page cache)
  parse     host marker walk + table building of ssg_amd.jpeg (no device work)
  decode    ssg_amd.jpeg.decode_batch: upload + Huffman / IDCT / colour kernels, device-synchronised
  loader    the whole GpuBatchLoader iteration (read + decode + resize + normalise), images/s -- the figure to hold against the embedder's
Prints one JSON line.  --host-only: only the stages that need no GPU (read, parse)."""
import argparse
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def picture(rng, h, w):
    """a person-crop-like picture: smooth colour regions + texture + sensor noise (a noise image would be 3x the size of a real file)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        img[..., c] = 110 + 60 * np.sin(xx / rng.uniform(6, 30) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(8, 40) + rng.uniform(0, 6))
    for _ in range(6):                       # blocks of clothing colour
        y0, x0 = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
        img[y0:y0 + int(rng.integers(8, h // 2)), x0:x0 + int(rng.integers(8, w // 2))] += rng.uniform(-70, 70, 3)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_files(root, n, restart, seed=1, distinct=512):
    from PIL import Image
    rng = np.random.default_rng(seed)
    blobs = []
    for k in range(min(n, distinct)):
        b = io.BytesIO()
        kw = dict(quality=int(rng.integers(80, 96)))
        if restart:
            kw["restart_marker_blocks"] = restart
        Image.fromarray(picture(rng, 128, 64)).save(b, "JPEG", **kw)
        blobs.append(b.getvalue())
    names = []
    for k in range(n):
        name = "%05d_c1_%06d.jpg" % (k % 751, k)
        with open(os.path.join(root, name), "wb") as f:
            f.write(blobs[k % len(blobs)])
        names.append((name, k % 751, 0))
    return names, float(np.mean([len(b) for b in blobs]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12936)
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--host-only", action="store_true")
    ap.add_argument("--restart", type=int, default=2, help="restart interval (MCU rows) of the second file set")
    a = ap.parse_args()
    import ssg_amd
    from ssg_amd import jpeg as pj
    out = {"n": a.n, "batch": a.batch}
    for label, restart in (("no_restart_markers", 0), ("restart_markers", a.restart)):
        with tempfile.TemporaryDirectory() as root:
            names, avg = make_files(root, a.n, restart)
            r = {"avg_file_bytes": round(avg)}
            t0 = time.perf_counter()
            blobs = [open(os.path.join(root, nm), "rb").read() for nm, _, _ in names]
            r["read_files_per_s"] = round(a.n / (time.perf_counter() - t0))
            t0 = time.perf_counter()
            for i in range(0, a.n, a.batch):
                pj.parse_batch(blobs[i:i + a.batch])
            r["parse_files_per_s"] = round(a.n / (time.perf_counter() - t0))
            if not a.host_only:
                import torch
                dev = torch.device("cuda", 0)
                pj.decode_batch(blobs[:a.batch], dev); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(0, a.n, a.batch):
                    pj.decode_batch(blobs[i:i + a.batch], dev)
                torch.cuda.synchronize()
                r["decode_files_per_s"] = round(a.n / (time.perf_counter() - t0))
                # device time of the three decode kernels alone (HIP events around the C-ABI call)
                from ssg_amd import _lib
                real = _lib.lib(); evs = []

                class P:
                    def __getattr__(self, k):
                        fn = getattr(real, k)
                        if k != "ssg_jpeg_decode_batch":
                            return fn

                        def timed(*args):
                            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                            e0.record(); rc = fn(*args); e1.record(); evs.append((e0, e1)); return rc
                        return timed
                _lib._lib = P()
                for i in range(0, a.n, a.batch):
                    pj.decode_batch(blobs[i:i + a.batch], dev)
                torch.cuda.synchronize(); _lib._lib = real
                r["decode_kernels_files_per_s"] = round(a.n / (sum(x.elapsed_time(y) for x, y in evs) * 1e-3))
                for mode in ("gpu", "pillow"):
                    n_eff = a.n if mode == "gpu" else min(a.n, 3000)
                    ld = ssg_amd.GpuBatchLoader(names[:n_eff], root=root, height=256, width=128, batch_size=a.batch, decode=mode)
                    for _ in ssg_amd.GpuBatchLoader(names[:a.batch], root=root, height=256, width=128, batch_size=a.batch, decode=mode):
                        pass
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in ld:
                        pass
                    torch.cuda.synchronize()
                    r["loader_%s_images_per_s" % mode] = round(n_eff / (time.perf_counter() - t0))
            out[label] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
