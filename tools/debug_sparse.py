import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from synth import clustered, hard_clustered
from ssg_amd import rerank, cluster
dev = torch.device("cuda", 0)
for name, tgt, src, lam in (("ragged", clustered(1531, 64, 7), clustered(400, 64, 8, intra=0.7), 0.3), ("n1536", clustered(1536, 64, 7), clustered(400, 64, 8, intra=0.7), 0.3),
                            ("hard6000", hard_clustered(6000, 128, 11), hard_clustered(2000, 128, 12, intra=0.7), 0.3)):
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=lam, validate=False)
    sp = h.sparse
    torch.cuda.synchronize()
    print(name, "cursor", sp["cursor"].tolist(), "cap", sp["cap"], "seg_len sum/max/min", int(sp["seg_len"].sum()), int(sp["seg_len"].max()), int(sp["seg_len"].min()),
          "seg_off min/max", int(sp["seg_off"].min()), int(sp["seg_off"].max()), "pending", h._pending.tolist())
    h.validate(); print("  sparse_ok", h.sparse_ok)
    # the eps rule's compaction through S: did the device-side gate take the sparse walk?
    from ssg_amd import _lib
    from ssg_amd._lib import check, ptr, stream
    L = _lib.lib(); st = stream()
    args = (ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.mode, h.lambda_value)
    hist = torch.zeros(2 * 4097, dtype=torch.int64, device=dev); thr3 = torch.zeros(5, dtype=torch.int64, device=dev)
    stride = max(1, h.nrows // 192)
    check(L.ssg_eps_sample_hist(*args, stride, None, ptr(hist[:4097]), st), "h1")
    check(L.ssg_eps_select_threshold(ptr(hist[:4097]), 1.3 * 1.6e-3, ptr(thr3), st), "sel")
    check(L.ssg_eps_sample_hist(*args, stride, ptr(thr3), ptr(hist[4097:]), st), "h2")
    check(L.ssg_eps_refine_threshold(ptr(hist[4097:]), ptr(thr3), st), "ref")
    buf = torch.empty(1 << 22, dtype=torch.int64, device=dev); cur = torch.zeros(3, dtype=torch.int64, device=dev)
    check(L.ssg_eps_compact_below_s(ptr(h.M), ptr(h.v), h.N, h.row0, h.nrows, h.lambda_value, ptr(thr3), ptr(buf), 1 << 22, ptr(cur), ptr(sp["pool"]),
                                    ptr(sp["seg_off"]), ptr(sp["seg_len"]), sp["nseg"], ptr(sp["cursor"]), ptr(sp["vmin"]), sp["jp0"], ptr(sp["rowmask"]), st), "cs")
    print("  rows left to the dense pass:", int(sp["rowmask"].sum().item()), "of", h.nrows, "vmin", float(np.uint16(int(sp["vmin"].item())).view(np.float16)))
    cur2 = torch.zeros(2, dtype=torch.int64, device=dev)
    check(L.ssg_eps_compact_below(*args, ptr(thr3), ptr(buf), 1 << 22, ptr(cur2), st), "cd")
    thr = float(np.uint32(int(thr3[0].item()) & 0xFFFFFFFF).view(np.float32))
    print("  thr", thr, "jp0", float(np.uint16(sp["jp0"]).view(np.float16)), "sparse cursor3", cur.tolist(), "dense cursor2", cur2.tolist())
