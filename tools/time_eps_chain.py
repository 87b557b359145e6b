#!/usr/bin/env python3
"""eps rule + DBSCAN chain (cluster.eps_rule_dbscan) at the bench shape (development aid): GPU time between HIP events with the launch queue
kept full (as in the bench step, where the host runs ahead of the re-rank's kernels), wall time of an isolated call, and the host-side
cost of queueing the chain.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from ssg_amd import rerank, cluster
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 16000)); Ns = 12936 * N // 16000
src = torch.from_numpy(synth.hard_clustered(Ns, 2048, 2, intra=0.7)).to(dev); tgt = torch.from_numpy(synth.hard_clustered(N, 2048, 1)).to(dev)
h = rerank.re_ranking_device(src, tgt, k1=20, k2=6, lambda_value=0.3, keep_euclid=False, validate=False)
pend0 = h._pending.clone() if h._pending is not None else None
ref = cluster.eps_rule_dbscan(h, 1.6e-3, min_samples=4)
# PIECES=1: every timed call finds the re-rank's status words still unread, as the bench's step does (they travel with the chain's read)
PIECES = os.environ.get("PIECES", "0") == "1"
_orig = cluster.eps_rule_dbscan
def _with_pieces(hh, *a, **kw):
    if PIECES and pend0 is not None:
        hh._pending = pend0
    return _orig(hh, *a, **kw)
cluster.eps_rule_dbscan = _with_pieces
R = 20
# (a) queue kept full: a long filler kernel in front, then the chain; events around the chain only
filler = torch.empty(1 << 28, dtype=torch.float32, device=dev)
gpu = []
for _ in range(R):
    filler.fill_(1.0); filler.mul_(2.0)        # ~2 x 0.3 ms of GPU work the host does not wait for
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = cluster.eps_rule_dbscan(h, 1.6e-3, min_samples=4); e1.record()
    torch.cuda.synchronize(); gpu.append(e0.elapsed_time(e1))
# (b) isolated call, wall clock
wall = []
for _ in range(R):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = cluster.eps_rule_dbscan(h, 1.6e-3, min_samples=4)
    wall.append((time.perf_counter() - t0) * 1e3)
assert out[0] == ref[0] and (out[3] == ref[3]).all()
gpu.sort(); wall.sort()
print("eps rule + DBSCAN chain, N=%d, SSG_EPS_SORT=%s: GPU time behind a full queue median %.3f ms (min %.3f); isolated call wall median %.3f ms (min %.3f); "
      "eps %.6f, %d clusters" % (N, os.environ.get("SSG_EPS_SORT", "sample"), gpu[R // 2], gpu[0], wall[R // 2], wall[0], out[0], int(out[3].max()) + 1))
