#!/usr/bin/env python3
"""Interleaved A/B of library builds / env knobs inside ONE process (development aid): every repetition runs one embedding forward per
variant, round-robin, with HIP events around every C-ABI launch -- run-to-run and box-to-box noise (+-3 %, DVFS) is common to the variants.

usage: ab_inproc.py [--B 1000] [--reps 6] name=lib.so[,ENV=VAL,...] ...
Each variant gets its own dlopen'ed copy of the library (its `static` knob caches are read under its own environment at the first call)."""
import argparse
import ctypes
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--json", default="")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import ssg_amd
    from ssg_amd import _lib
    import layer_table as lt
    dev = torch.device("cuda", 0)
    protos = _lib.parse_header()
    tmp = tempfile.mkdtemp(prefix="ab_")
    m = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, pretrained=False).cuda().eval()
    x = torch.randn(a.B, 3, 256, 128, device=dev)
    var = []
    for i, spec in enumerate(a.variants):
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        path, envs = parts[0], dict(p.split("=", 1) for p in parts[1:])
        cp = os.path.join(tmp, "lib%d.so" % i)
        shutil.copy(os.path.join(ROOT, path) if not os.path.isabs(path) else path, cp)
        old = {k: os.environ.get(k) for k in envs}
        os.environ.update(envs)
        L = ctypes.CDLL(cp)
        for fn, (res, args) in protos.items():
            f = getattr(L, fn); f.restype = res; f.argtypes = args
        _lib._lib = L
        m._fmap(x); torch.cuda.synchronize()          # knob caches are filled under this variant's environment
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        var.append(dict(name=name, L=L, sums=None, tot=[]))
    rows0 = None
    for rep in range(a.reps):
        for v in (var if rep % 2 == 0 else var[::-1]):
            rec = lt.Rec(v["L"]); _lib._lib = rec
            m._fmap(x); torch.cuda.synchronize()
            ms = [r[2].elapsed_time(r[3]) for r in rec.rows]
            v["sums"] = ms if v["sums"] is None else [p + q for p, q in zip(v["sums"], ms)]
            v["tot"].append(sum(ms))
            rows0 = rows0 or rec.rows
    print("%-50s" % "launch" + "".join("%10s" % v["name"] for v in var))
    for i, r in enumerate(rows0):
        fl, by, label = lt.work(r[0], r[1], a.B)
        if fl == 0:
            continue
        print("%2d %-47s" % (i, label) + "".join("%10.3f" % (v["sums"][i] / a.reps) for v in var))
    print("%-50s" % "total ms (mean)" + "".join("%10.3f" % (sum(v["tot"]) / a.reps) for v in var))
    print("%-50s" % "total ms (min)" + "".join("%10.3f" % min(v["tot"]) for v in var))
    print("%-50s" % "total ms (median)" + "".join("%10.3f" % sorted(v["tot"])[len(v["tot"]) // 2] for v in var))
    if a.json:
        import json
        json.dump({v["name"]: dict(per_launch=[s / a.reps for s in v["sums"]], totals=v["tot"]) for v in var}, open(a.json, "w"))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
