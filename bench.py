#!/usr/bin/env python3
"""bench.py -- one SSG grouping iteration per step on N GPUs of one node.

Step (BASELINE.json configs[1]+[2], N=16 000): embed the source (12 936) and target (16 000)
sets with ResNet-50 (256x128 synthetic images, original + flipped forward, L2 norm) -> full
k-reciprocal re-rank distance (k1=20, k2=6, lambda=0.3) incl. the source term -> eps rule ->
DBSCAN, all through the HIP kernels behind include/ssg_hip.h.  Inputs are resident in HBM when
the timed region starts.  With N>1 ranks (torchrun) the images and the N x N row blocks are
sharded and the embeddings / small tables all-gathered over RCCL (strong scaling, N fixed).

Two synthetic tracks: random-init backbone features are degenerate (the
reference NaNs on them, reid/rerank.py:40), so the grouping leg consumes clustered unit-norm
embeddings of the same shape that are also HBM-resident; both legs are inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fp32-MFMA implicit
GEMM convolution); `roofline_kernels` lists the HBM-bound distance / re-rank / DBSCAN kernels
against their algorithmic bytes; `cpu_baseline` times the CPU oracle (a port
of the reference algorithm) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from roofline import (KernelTimer, grouping_roofline, PEAK_FP32_MFMA_TF, PEAK_FP16_MFMA_TF, PEAK_HBM_GBS)  # noqa: E402,F401  (shared with tools/run_configs.py)
FLOP_PER_IMAGE = 10.68e9      # 2 forwards x 2 x 2.669 GMAC


def build_fingerprint():
    """sha256 of the kernel sources the HIP library is built from (csrc/*.hip, *.h, include/ssg_hip.h): ties a PMC traffic summary
    under profiles/ to the build it was taken on -- bench.py refuses a summary of another build instead of quoting stale bytes"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "self-similarity-grouping_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ssg_hip.h"), "rb").read())
    return h.hexdigest()[:16]


EMBED_SOURCES = ("conv.hip", "bottleneck.hip", "stem_pool.hip", "ssg_common.h")


def embed_fingerprint():
    """sha256 over the sources of the embedding's convolution kernels only: what a PMC traffic summary of the embedding depends on (a
    change to the grouping kernels does not make it stale)"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "self-similarity-grouping_amd", "csrc")
    for f in EMBED_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_conv_traffic.json")


def pmc_traffic(path, build, batch, launches_per_forward):
    """HBM bytes per convolution launch from the PMC summary `tools/pmc_embed.sh` wrote (FETCH_SIZE x 2 + WRITE_SIZE over one forward, divided
    by its launches) -> (bytes or None, note).  None when there is no summary, when it was taken on another build of the embedding kernels
    (`embed_build`, the fingerprint of `embed_fingerprint()`, differs) or on another launch set / batch size: a stale figure is not quoted."""
    try:
        pm = json.load(open(path))
        have = pm.get("embed_build", pm.get("build"))
        if have != build:
            return None, "%s was taken on another build (%s, this one is %s): not quoted" % (os.path.basename(path), have, build)
        if pm["batch"] != batch or pm["launches_per_forward"] != launches_per_forward:
            return None, "%s covers %d launches per forward at batch %d, this run has %d at batch %d: not quoted" % (
                os.path.basename(path), pm["launches_per_forward"], pm["batch"], launches_per_forward, batch)
        per = (pm["fetch_bytes_per_forward"] + pm["write_bytes_per_forward"]) / pm["launches_per_forward"]
        return round(per), ("HBM bytes per conv launch (average over the %d launches of a forward, batch %d) from %s; algorithmic %.0f"
                            % (pm["launches_per_forward"], pm["batch"], pm["source"], pm["algorithmic_bytes_per_forward"] / pm["launches_per_forward"]))
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


class SyncCounter:
    """counts the blocking device -> host reads (Tensor.item / tolist / cpu / numpy on a CUDA tensor) inside a `with` block: the host
    round trips of the grouping leg"""

    def __init__(self):
        self.n = 0

    def __enter__(self):
        self._orig = {k: getattr(torch.Tensor, k) for k in ("item", "tolist", "cpu")}
        me = self
        # (round 6: the one-GPU eps rule + DBSCAN chain reads back through DMA copies into page-locked memory + ONE stream wait: counted too)
        self._ssync = torch.cuda.Stream.synchronize

        def ssync(st, *a, **kw):
            me.n += 1
            return me._ssync(st, *a, **kw)
        torch.cuda.Stream.synchronize = ssync

        def wrap(name):
            fn = self._orig[name]

            def counted(t, *a, **kw):
                if t.is_cuda:
                    me.n += 1
                return fn(t, *a, **kw)
            return counted
        for k in self._orig:
            setattr(torch.Tensor, k, wrap(k))
        return self

    def __exit__(self, *exc):
        for k, fn in self._orig.items():
            setattr(torch.Tensor, k, fn)
        torch.cuda.Stream.synchronize = self._ssync


class CollectiveCounter:
    """counts the torch.distributed collectives (and the bytes this rank contributes / receives) issued inside a `with` block: the
    exchange steps of one sharded grouping leg.  World 1 issues none."""

    NAMES = ("all_gather_into_tensor", "all_gather", "all_reduce", "all_gather_object", "broadcast", "barrier")

    def __init__(self):
        self.calls, self.bytes_out, self.bytes_in = {}, 0, 0

    def __enter__(self):
        import torch.distributed as tdist
        self._d, self._orig = tdist, {k: getattr(tdist, k) for k in self.NAMES}
        me = self

        def wrap(name):
            fn = self._orig[name]

            def counted(*a, **kw):
                me.calls[name] = me.calls.get(name, 0) + 1
                if name == "all_gather_into_tensor":
                    me.bytes_in += a[0].numel() * a[0].element_size(); me.bytes_out += a[1].numel() * a[1].element_size()
                elif name == "all_gather":
                    me.bytes_in += sum(t.numel() * t.element_size() for t in a[0]); me.bytes_out += a[1].numel() * a[1].element_size()
                elif name == "all_reduce":
                    me.bytes_in += a[0].numel() * a[0].element_size(); me.bytes_out += a[0].numel() * a[0].element_size()
                return fn(*a, **kw)
            return counted
        for k in self.NAMES:
            setattr(tdist, k, wrap(k))
        return self

    def __exit__(self, *exc):
        for k, fn in self._orig.items():
            setattr(self._d, k, fn)

    def summary(self):
        return {"calls": dict(self.calls), "n": sum(self.calls.values()), "bytes_sent": self.bytes_out, "bytes_received": self.bytes_in}


def cpu_baseline(args, src, tgt, gpu_labels, gpu_eps):
    """Oracle ("port" of the reference algorithm, oracle/ssg_oracle.c + oracle/embed_oracle.py) on this box's host cores.
    The grouping leg is MEASURED at the bench size on the bench's own embeddings (and its labels are compared with the
    GPU's); the embedding leg on a bounded sample of images.  The thread counts are probed (8-image / N=1000 warm-ups):
    on the GPU boxes the full `sched_getaffinity` count oversubscribes badly (torch conv: 0.4 img/s with 256 threads,
    21 img/s with 32)."""
    from oracle import ssg_oracle as ora, embed_oracle
    import ssg_amd
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cand = sorted({c for c in (8, 16, 32, 64, 128) if c <= avail} | ({avail} if avail <= 128 else set()))
    sd = ssg_amd.synthetic_state_dict(seed=1)
    imgs = torch.randn(args.cpu_images, 3, 256, 128, generator=torch.Generator().manual_seed(1))
    best_t, best_r = cand[0], 0.0
    for c in [c for c in cand if c <= 64]:           # embed probe: 8 images after a 2-image warm-up
        torch.set_num_threads(c)
        embed_oracle.embed_with_flip(sd, imgs[:2], 1)
        t0 = time.time(); embed_oracle.embed_with_flip(sd, imgs[:8], 1); r = 8 / (time.time() - t0)
        if r > best_r:
            best_t, best_r = c, r
    torch.set_num_threads(best_t)
    t0 = time.time(); embed_oracle.embed_with_flip(sd, imgs, 1); t_embed = time.time() - t0
    best_o, best_s = cand[0], float("inf")
    for c in cand:                                   # grouping probe: N = 1000
        ora.set_num_threads(c)
        t0 = time.time(); ora.re_ranking(src[:1000], tgt[:1000], k1=20, k2=6, lambda_value=args.lambda_value); dt = time.time() - t0
        if dt < best_s:
            best_o, best_s = c, dt
    ora.set_num_threads(best_o)
    n = min(args.cpu_n, tgt.shape[0]) if args.cpu_n > 0 else tgt.shape[0]
    ns = src.shape[0] if n == tgt.shape[0] else min(n, src.shape[0])
    t0 = time.time()
    _, final = ora.re_ranking(src[:ns], tgt[:n], k1=20, k2=6, lambda_value=args.lambda_value)
    t_rr = time.time() - t0
    t0 = time.time()
    eps, _, _ = ora.eps_rule(final, args.rho); lab = ora.dbscan(final, eps, 4)
    t_cl = time.time() - t0
    del final
    img_s = args.cpu_images / t_embed
    # SURVEY.md 8d also asks for the reference's own CPU-runnable size, BASELINE configs[0]: N = 2 000 (no re-rank: pairwise L2 + eps + DBSCAN;
    # and the full re-rank for comparison), measured
    n2k = min(2000, tgt.shape[0]); ns2k = min(2000, src.shape[0])
    t0 = time.time(); e2k, _ = ora.re_ranking(src[:ns2k], tgt[:n2k], no_rerank=True); ee, _, _ = ora.eps_rule(e2k, args.rho); ora.dbscan(e2k, ee, 4); t_2k_plain = time.time() - t0
    t0 = time.time(); _, f2k = ora.re_ranking(src[:ns2k], tgt[:n2k], k1=20, k2=6, lambda_value=args.lambda_value); ee, _, _ = ora.eps_rule(f2k, args.rho); ora.dbscan(f2k, ee, 4)
    t_2k_rr = time.time() - t0
    del e2k, f2k
    full = n == args.N and ns == args.Ns
    scale = 1.0 if full else (args.N * (args.N + args.Ns)) / float(n * (n + ns))      # N*(N+Ns)*d cost model only when a smaller N was asked for
    est_iter = (args.N + args.Ns) / img_s + (t_rr + t_cl) * scale
    out = {"value": round((args.N + args.Ns) / est_iter, 3), "unit": "images/s", "cores": max(best_t, best_o), "kind": "port",
           "sample": "embed: %d images 256x128 incl. flip, torch fp32 on %d threads (best of %r; %.2f img/s measured); grouping: oracle "
                     "re_ranking+eps+DBSCAN MEASURED at N=%d Ns=%d d=2048 on %d OpenMP threads (best of %r): %.2f s%s; %d cores visible"
                     % (args.cpu_images, best_t, [c for c in cand if c <= 64], img_s, n, ns, best_o, cand, t_rr + t_cl,
                        "" if full else " (extrapolated to N=%d,Ns=%d by N*(N+Ns))" % (args.N, args.Ns), avail),
           "embed_images_per_s": round(img_s, 3), "rerank_s_measured": round(t_rr, 3), "eps_dbscan_s_measured": round(t_cl, 3),
           "rerank_dbscan_s_measured": round(t_rr + t_cl, 3), "rerank_dbscan_N": n, "rerank_dbscan_Ns": ns, "grouping_extrapolated": not full,
           "threads_embed": best_t, "threads_grouping": best_o, "cores_visible": avail,
           "n2000": {"what": "BASELINE configs[0] size on the same threads: N=%d, Ns=%d, d=2048 grouping only (the embedding of 2 000 + 2 000 images at the rate above: %.0f s)"
                             % (n2k, ns2k, (n2k + ns2k) / img_s),
                     "l2_eps_dbscan_s": round(t_2k_plain, 3), "rerank_eps_dbscan_s": round(t_2k_rr, 3)}}
    if full:
        out["labels_equal_gpu"] = bool(np.array_equal(lab, gpu_labels)) and float(eps) == float(gpu_eps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=16000)
    ap.add_argument("--Ns", type=int, default=12936)
    ap.add_argument("--batch", type=int, default=1000)     # images per embedding call (1000 x 64 x 32 x 256 x 4 B stays under the 2 GiB buffer range)
    ap.add_argument("--lambda_value", type=float, default=0.3)
    ap.add_argument("--rho", type=float, default=1.6e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=256)
    ap.add_argument("--cpu-n", type=int, default=0, help="grouping size of the CPU baseline (0 = the bench size, measured)")
    ap.add_argument("--track-g", choices=("hard", "separable"), default="hard",
                    help="embeddings of the timed grouping leg: 'hard' (tools/synth.hard_clustered: noise, border points, unequal identity "
                         "sizes) or 'separable' (SURVEY 8d: 16 per identity, trivially separable); the other one is timed once, untimed-region")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (f32 embed, stable rank mode, other track)")
    ap.add_argument("--grouping", choices=("auto", "shard", "replicate"), default="auto",
                    help="N > 1: form of the grouping leg -- 'shard' (row blocks + all-gathers), 'replicate' (every rank runs the whole leg, no "
                         "collective) or 'auto' (ssg_amd.dist.choose_grouping: the form its time model predicts to be faster for this N and world size)")
    ap.add_argument("--uniform", action="store_true",
                    help="time the PRODUCT DEFAULT only: no per-launch HIP events anywhere in the timed region (every embedding batch on two streams, "
                         "the re-rank's source term on its second stream on every step); the roofline objects of the line are then null -- a cross-check "
                         "of `value` against the default run, whose timed steps alternate two instrumentation modes")
    args = ap.parse_args()

    import ssg_amd
    from ssg_amd import _lib, cluster, dist as sdist, evaluators, rerank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SSG_BENCH_SHARE_GPU") == "1":           # dry run of the N > 1 code path on a one-GPU box: every rank on cuda:0, gloo collectives
        local = 0
    torch.cuda.set_device(local)                               # before the process group: RCCL binds its communicator to the current device
    rank, world, group = sdist.init_from_env(backend="gloo" if os.environ.get("SSG_BENCH_SHARE_GPU") == "1" else None)
    if world != args.gpus:
        raise SystemExit("launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d, --gpus %d)" % (args.gpus, world, args.gpus))
    dev = torch.device("cuda", local)
    timer = KernelTimer(_lib.lib())
    _lib._lib = timer

    # ---- inputs resident in HBM before the timed region.  Every rank generates and holds ONLY its own block of the synthetic image
    # sets (the `shard_bounds` block the product API `extract_embeddings(..., group=)` hands it: 11.4 GB / world; seed 1 + rank, so one
    # GPU sees the same images as before) -- start-up work and memory no longer grow with the number of ranks
    model = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False).cuda(local).eval()
    precision = model.precision
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    t_lo, t_hi = sdist.shard_bounds(args.N, rank, world)
    s_lo, s_hi = sdist.shard_bounds(args.Ns, rank, world)
    tgt_imgs = torch.randn(t_hi - t_lo, 3, 256, 128, generator=g, device=dev)
    src_imgs = torch.randn(s_hi - s_lo, 3, 256, 128, generator=g, device=dev)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    gens = {"hard": synth.hard_clustered, "separable": synth.clustered}
    emb_np = {k: (g(args.Ns, 2048, 2, intra=0.7), g(args.N, 2048, 1)) for k, g in gens.items() if k == args.track_g or not args.no_extras}
    src_emb = torch.from_numpy(emb_np[args.track_g][0]).to(dev)
    tgt_emb = torch.from_numpy(emb_np[args.track_g][1]).to(dev)
    # N > 1: the grouping leg runs in the form the product's compute_dist(..., group=, grouping='auto') would choose (VERDICT r5 #5a)
    grouping_form = sdist.choose_grouping(args.N, world, args.grouping)
    g_group = group if grouping_form == "shard" else None
    row0, row1 = sdist.shard_bounds(args.N, rank, world) if g_group is not None else (0, args.N)      # ragged row blocks (N need not divide by the number of GPUs)
    nrows = row1 - row0

    class TimedLoader(evaluators.TensorBatchLoader):
        """the product's resident-tensor loader; per-launch HIP events are switched on for every 8th batch only"""

        def shard(self, r, w):
            s_ = super().shard(r, w)
            return TimedLoader(s_.images, s_.batch_size, s_.fnames, s_.pids, s_.first, s_.count, s_.base)

        def __iter__(self):
            for bi, batch in enumerate(super().__iter__()):
                timer.sample = (bi % 8 == 0)
                # sampled batches run their two forwards one after the other on one stream, so that an event pair brackets ONE launch
                # running alone (the roofline figure); the other batches use the product default, two streams (launches overlap)
                model.flip_streams = not (timer.on and timer.sample)
                if timer.on and timer.sample:
                    timer.sampled_images += batch[0].shape[0]
                yield batch
            timer.sample = True; model.flip_streams = True

    # the loaders describe the WHOLE sets (count = N / Ns: block lengths and names of every rank follow from it); a rank's resident block
    # starts at item `base`
    tgt_loader = TimedLoader(tgt_imgs, args.batch, count=args.N, base=t_lo)
    src_loader = TimedLoader(src_imgs, args.batch, count=args.Ns, base=s_lo)
    imgs_rank = tgt_loader.shard(rank, world).num_items() + src_loader.shard(rank, world).num_items()

    def step(group_events=True):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        # image batches sharded by rank, C1 = all-gather of the embeddings over xGMI -- the product function
        f_src, _, _ = evaluators.extract_embeddings(model, src_loader, group=group)
        f_tgt, _, _ = evaluators.extract_embeddings(model, tgt_loader, group=group)
        ev[1].record()
        assert f_tgt.shape == (args.N, 2048) and f_src.shape == (args.Ns, 2048)
        # the per-launch HIP events of the grouping leg (two event records around each of its ~45 launches) are themselves ~0.2 ms of its
        # ~6 ms: they are recorded on every other timed step only, and rerank_ms / eps_dbscan_ms are read from the steps WITHOUT them
        timer.sample = group_events
        # ... and on those steps the re-rank keeps to ONE stream (SSG_RERANK_OVERLAP=0), so that an event pair brackets a launch that runs
        # alone -- the product default (the source term on a second stream beside the k-reciprocal kernels) is what the other steps time
        os.environ["SSG_RERANK_OVERLAP"] = "0" if (group_events and timer.on) else "1"
        h = rerank.re_ranking_device(src_emb, tgt_emb, k1=20, k2=6, lambda_value=args.lambda_value, keep_euclid=False, validate=False,
                                     row0=row0, nrows=nrows, group=g_group)
        ev[2].record()
        # eps rule + DBSCAN as the product's generate_selflabel runs them at iteration 0 (selftraining.py:289-306): one device chain, one read
        eps, cnt, top, labels, _ = cluster.eps_rule_dbscan(h, args.rho, min_samples=4)
        ev[3].record()
        timer.sample = True
        os.environ.pop("SSG_RERANK_OVERLAP", None)
        return ev, eps, labels

    def sync_barrier():
        torch.cuda.synchronize()
        sdist.barrier(group)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_barrier()
    timer.on = os.environ.get("SSG_BENCH_NO_KERNEL_EVENTS", "0") != "1" and not args.uniform   # (--uniform: no per-launch events at all)
    legs = []
    with_events = [si % 2 == 0 for si in range(args.steps)]        # steps 0, 2, 4, ..: grouping launches bracketed by events (roofline figures)
    t0 = time.perf_counter()
    for si in range(args.steps):
        ev, eps, labels = step(with_events[si])
        legs.append(ev)
    sync_barrier()
    elapsed = time.perf_counter() - t0
    timer_was_on = timer.on
    timer.on = False
    if group is not None:
        import torch.distributed as tdist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    ms_step = elapsed * 1e3 / args.steps
    t_embed = sum(e[0].elapsed_time(e[1]) for e in legs) / args.steps
    n_ev_steps = sum(with_events) if timer_was_on else 0
    clean = [e for e, w in zip(legs, with_events) if not (w and timer_was_on)] or legs      # (--steps 1: the one step carries the events)
    t_rerank = sum(e[1].elapsed_time(e[2]) for e in clean) / len(clean)
    t_cluster = sum(e[2].elapsed_time(e[3]) for e in clean) / len(clean)
    evd = [e for e, w in zip(legs, with_events) if w and timer_was_on]
    grouping_timing = {"steps_without_kernel_events": len(clean) if clean is not legs else 0, "steps_with_kernel_events": len(evd),
                       "rerank_ms_with_kernel_events": round(sum(e[1].elapsed_time(e[2]) for e in evd) / len(evd), 3) if evd else None,
                       "eps_dbscan_ms_with_kernel_events": round(sum(e[2].elapsed_time(e[3]) for e in evd) / len(evd), 3) if evd else None,
                       "what": "rerank_ms / eps_dbscan_ms = HIP-event time of the leg on the timed steps whose launches are NOT bracketed by per-launch "
                               "events (product default: the source term overlaps the k-reciprocal kernels on a second stream); the per-kernel roofline "
                               "figures come from the other timed steps, where every launch runs alone on one stream (the event records cost ~0.1 ms per leg)"}
    tot = timer.totals()

    # ---- untimed extras (rank 0, single GPU): what the headline configuration costs relative to its alternatives
    extras = {}
    if world == 1 and not args.no_extras:
        def timed_ms(fn, reps=2):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / reps
        # (a) initial ranking: the reference's introsort tie order (default) vs the canonical stable order
        D, rowmax, _ = rerank._original_distance(_lib.lib(), tgt_emb, 0, args.N, float(tgt_emb.abs().max()), _lib.stream())
        t_intro = timed_ms(lambda: rerank.initial_rank(D, rowmax, args.N, args.N, 21, "introsort"))
        t_stable = timed_ms(lambda: rerank.initial_rank(D, rowmax, args.N, args.N, 21, "stable"))
        del D, rowmax
        t_rr_stable = timed_ms(lambda: rerank.re_ranking_device(src_emb, tgt_emb, k1=20, k2=6, lambda_value=args.lambda_value, keep_euclid=False,
                                                                 rank_mode="stable"))
        extras["rank_mode"] = {"default": "introsort (np.argsort default kind replayed on the device: bit parity with the unmodified reference)",
                               "introsort_kernel_ms": round(t_intro, 3), "stable_kernel_ms": round(t_stable, 3),
                               "rerank_ms_with_stable_order": round(t_rr_stable, 3)}
        # (b) the embedding without the split-half trick: every product on the fp32 matrix cores
        m32 = ssg_amd.create("resnet50", num_classes=0, num_split=1, cluster=False, seed=1, pretrained=False, precision="f32").cuda(local).eval()
        n32 = min(4 * args.batch, tgt_imgs.shape[0])
        t32 = timed_ms(lambda: [m32.embed_with_flip(tgt_imgs[i:i + args.batch]) for i in range(0, n32, args.batch)], reps=1)
        extras["embed_precision_f32"] = {"images_per_s": round(n32 / (t32 * 1e-3), 1), "images": n32,
                                         "what": "same embedding with precision='f32' (v_mfma_f32_32x32x2 for every product), timed on %d images" % n32}
        del m32
        # (c) the grouping leg on the other synthetic track
        other = "separable" if args.track_g == "hard" else "hard"
        so, to = (torch.from_numpy(a).to(dev) for a in emb_np[other])
        def grouping_other():
            ho = rerank.re_ranking_device(so, to, k1=20, k2=6, lambda_value=args.lambda_value, keep_euclid=False, validate=False)
            e_, _, _, l_, _ = cluster.eps_rule_dbscan(ho, args.rho, min_samples=4)
            return e_, l_
        t_other = timed_ms(grouping_other, reps=2)
        e_o, l_o = grouping_other()
        extras["grouping_other_track"] = {"track": other, "rerank_dbscan_ms": round(t_other, 3), "clusters": int(l_o.max() + 1),
                                          "noise": int((l_o < 0).sum()), "eps": e_o}
        del so, to
        # (d) the LITERAL drop-in chain INTEGRATION.md section 1 tells a maintainer to call, next to the fused path the timed region runs
        # ("API materialisation timed and reported separately"): extract_features -> OrderedDict of N CPU tensors
        # (reid/evaluators.py:18-60) -> the reorder + stack loop of selftraining.py:197-209 -> compute_dist (CPU tensors in: re-upload)
        # -> generate_selflabel; and re_ranking's numpy return (rerank.py:27: float16 euclidean_dist + float64 final_dist on the host)
        import contextlib
        import io
        from types import SimpleNamespace
        from ssg_amd import selftraining
        names_t = ["t%06d.jpg" % i for i in range(args.N)]; names_s = ["s%06d.jpg" % i for i in range(args.Ns)]
        quiet = contextlib.redirect_stdout(io.StringIO())
        torch.cuda.synchronize(); c0 = time.perf_counter()
        tf, _ = evaluators.extract_features(model, evaluators.TensorBatchLoader(tgt_imgs, args.batch, names_t), print_freq=0, for_eval=False)
        sf, _ = evaluators.extract_features(model, evaluators.TensorBatchLoader(src_imgs, args.batch, names_s), print_freq=0, for_eval=False)
        c1 = time.perf_counter()
        target_features = torch.cat([tf[f].unsqueeze(0) for f in names_t], 0)
        source_features = torch.cat([sf[f].unsqueeze(0) for f in names_s], 0)
        c2 = time.perf_counter()
        assert target_features.shape == (args.N, 2048) and source_features.shape == (args.Ns, 2048) and not target_features.is_cuda
        del tf, sf, target_features, source_features
        # (as for the numpy return below: the first call page-locks its destination -- 0.7 GB here --, a loop that drops the previous
        # iteration's dictionaries re-uses the pool: the second call is the steady state of selftraining.py's iterations)
        import gc
        gc.collect()
        torch.cuda.synchronize(); c0b = time.perf_counter()
        tf, _ = evaluators.extract_features(model, evaluators.TensorBatchLoader(tgt_imgs, args.batch, names_t), print_freq=0, for_eval=False)
        sf, _ = evaluators.extract_features(model, evaluators.TensorBatchLoader(src_imgs, args.batch, names_s), print_freq=0, for_eval=False)
        c1b = time.perf_counter()
        del tf, sf
        # (the grouping calls get the clustered track as CPU tensors -- the form the reference holds its stacked features in; the
        # embedder's own output on N(0,1) images is degenerate, see `legs`)
        src_cpu, tgt_cpu = torch.from_numpy(emb_np[args.track_g][0]), torch.from_numpy(emb_np[args.track_g][1])
        ns_args = SimpleNamespace(no_rerank=False, rho=args.rho)
        with quiet:
            selftraining.generate_selflabel(*selftraining.compute_dist(src_cpu, tgt_cpu, lambda_value=args.lambda_value, no_rerank=False, num_split=1), 0, ns_args, [])
        torch.cuda.synchronize(); c3 = time.perf_counter()
        e_l, r_l = selftraining.compute_dist(src_cpu, tgt_cpu, lambda_value=args.lambda_value, no_rerank=False, num_split=1)
        torch.cuda.synchronize(); c4 = time.perf_counter()
        with quiet:
            lab_l, _ = selftraining.generate_selflabel(e_l, r_l, 0, ns_args, [])
        c5 = time.perf_counter()
        del e_l, r_l
        # (first call: the destination pages are page-locked -- hostio.py's pool -- ; a loop that drops the previous iteration's matrices
        # re-uses them: the second call is the steady state of selftraining.py's loop)
        with quiet:
            eu_np, fin_np = rerank.re_ranking(emb_np[args.track_g][0], emb_np[args.track_g][1], k1=20, k2=6, lambda_value=args.lambda_value)
        t_rr_first = time.perf_counter() - c5
        del eu_np, fin_np
        import gc
        gc.collect()
        c5b = time.perf_counter()
        with quiet:
            eu_np, fin_np = rerank.re_ranking(emb_np[args.track_g][0], emb_np[args.track_g][1], k1=20, k2=6, lambda_value=args.lambda_value)
        c6 = time.perf_counter()
        with quiet:        # the reference's own consumer of that matrix (selftraining.py:280-313): the array still carries its device handle
            lab_n, _ = selftraining.generate_selflabel([[]], [fin_np], 0, ns_args, [])
        c7 = time.perf_counter()
        extras["dropin_chain"] = {
            "what": "untimed-region cost of the literal drop-in surface (INTEGRATION.md section 1) beside the fused device path of the timed region, seconds",
            "extract_features_dicts_s": round(c1b - c0b, 4), "extract_features_dicts_first_call_s": round(c1 - c0, 4), "extract_images": args.N + args.Ns,
            "stack_loop_selftraining_197_209_s": round(c2 - c1, 4),
            "compute_dist_from_cpu_tensors_s": round(c4 - c3, 5), "generate_selflabel_s": round(c5 - c4, 5),
            "re_ranking_numpy_return_s": round(c6 - c5b, 4), "re_ranking_numpy_return_first_call_s": round(t_rr_first, 4),
            "re_ranking_numpy_return_bytes": int(eu_np.nbytes + fin_np.nbytes),
            "re_ranking_numpy_return_GBps": round((eu_np.nbytes + fin_np.nbytes) / max(c6 - c5b - t_rerank * 1e-3, 1e-9) / 1e9, 1),
            "re_ranking_numpy_return_what": "re_ranking() end to end incl. the H2D of the features and the device pipeline; _s = second call of the size (page-locked "
                                            "destination re-used from hostio's pool: the steady state of a self-training loop), first_call_s = incl. page-locking 2.56 GB",
            "generate_selflabel_on_numpy_final_dist_s": round(c7 - c6, 5),
            "labels_equal_fused": bool(np.array_equal(lab_l[0], labels)) and bool(np.array_equal(lab_n[0], labels)),
            "fused_path_for_comparison_s": {"embed": round(t_embed * 1e-3, 4), "rerank_eps_dbscan": round((t_rerank + t_cluster) * 1e-3, 5)}}
        del eu_np, fin_np

    # host round trips and collectives of one grouping leg (re-rank + eps rule + DBSCAN), counted on an extra untimed pass -- on every
    # rank, at any world size (the sharded path adds its own blocking reads: sizes of the ragged candidate / edge blocks)
    with CollectiveCounter() as cc, SyncCounter() as sc:
        h_ = rerank.re_ranking_device(src_emb, tgt_emb, k1=20, k2=6, lambda_value=args.lambda_value, keep_euclid=False, validate=False,
                                      row0=row0, nrows=nrows, group=g_group)
        n_rr = sc.n
        res_ = cluster.eps_rule_dbscan(h_, args.rho, min_samples=4)
        n_ed = sc.n - n_rr
    row_split = cluster.sparse_row_split(h_, args.rho, res_[0]) if g_group is None else None     # which path the rows of the sparse passes took (untimed)
    del h_
    host_syncs = {"rerank": n_rr, "eps_rule_dbscan": n_ed, "per_split": n_rr + n_ed, "world": world,
                  "what": "blocking device->host reads (item/tolist/cpu) of one grouping leg on rank 0: value ranges of the features (re-rank); ONE read for "
                          "eps rule + DBSCAN on one GPU (labels, neighbour counts, eps, check words, the re-rank's status words -- cluster.eps_rule_dbscan); "
                          "sharded rows (round 6) run the same chain with two all-gathers inside and the same ONE read (labels + counts + check words + every rank's status words); a failed check falls back to the two-call form"}
    collectives = cc.summary()
    collectives["what"] = ("torch.distributed collectives of one grouping leg on rank 0 (all-gathers of the row-block tables: source minima, rank lists, V, V_qe, "
                           "eps candidates, neighbour counts, edges; all-reduces of the eps histograms / counters); bytes = this rank's contribution / what it receives")
    if rank != 0:
        return
    n_img = args.N + args.Ns
    # ---- roofline of the dominant kernel: implicit-GEMM convolution on the fp32 matrix cores
    CONV_ABI = ("ssg_conv2d_nhwc_x", "ssg_conv1x1_dual_nhwc_x", "ssg_conv2d_nhwc_f32", "ssg_conv1x1_dual_nhwc_f32",
                "ssg_stem_pool_nchw_x", "ssg_bottleneck_nhwc_x", "ssg_bottleneck_ds_nhwc_x")     # every launch that carries convolution flops
    convs = [tot[k] for k in CONV_ABI if k in tot]
    n_conv, ms_conv = (sum(c[0] for c in convs), sum(c[1] for c in convs)) if convs else (1, float("nan"))
    n_conv_per_fwd = round(n_conv * args.batch / max(2 * timer.sampled_images, 1)) if convs else 0
    launches_by_abi = {k: {"launches": tot[k][0], "ms": round(tot[k][1], 3)} for k in CONV_ABI if k in tot}
    conv_tf = timer.sampled_images * FLOP_PER_IMAGE / (ms_conv * 1e-3) / 1e12     # launches and images of the sampled batches
    split = precision == "split"
    # split-half path: every fp32 multiply-add is three fp16-MFMA multiply-adds (xh*wh + xh*wl + xl*wh), so the
    # fp32-equivalent ceiling of the fp16 matrix cores is a third of their dense peak; frac = executed/peak either way
    peak = PEAK_FP16_MFMA_TF / 3.0 if split else PEAK_FP32_MFMA_TF
    roof = {"bound": "mfma",
            "kernel": ("the convolution stack of the embedding, split-half fp32 on v_mfma_f32_32x32x16_f16 (3 MFMA products per multiply, fp32 "
                       "accumulate): conv_dma_kernel / conv_igemm_kernel launches plus the fused stem (conv1 + bn + relu + maxpool) and the fused "
                       "layer1 bottleneck blocks; 53 convolutions x 2 orientations per image in %d launches per forward" % (n_conv_per_fwd,) if split else
                       "conv_igemm_kernel (fp32 v_mfma_f32_32x32x2, 53 convs x 2 orientations per image)"),
            "achieved": round(conv_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(conv_tf / peak, 4),
            "traffic": None, "launches": n_conv, "avg_launch_ms": round(ms_conv / max(n_conv, 1), 4), "launches_by_abi": launches_by_abi,
            "algorithmic": "10.68 GFLOP per image (2 forwards x 5.34 GFLOP); HIP events around every convolution-carrying launch of every 8th batch (those batches run their two forwards on one stream): %d of the "
                           "%d images embedded in the timed steps" % (timer.sampled_images, imgs_rank * args.steps)}
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, collected on this
    # kernel set at the same batch size); null when the configuration differs from the profiled one
    if split:
        roof["traffic"], note = pmc_traffic(PMC_TRAFFIC_JSON, embed_fingerprint(), args.batch, n_conv_per_fwd)
        if note:
            roof["traffic_note"] = note
    if split:
        roof["peak_is"] = "fp16 dense MFMA peak %.1f / 3 products per fp32 multiply" % PEAK_FP16_MFMA_TF
        roof["executed_fp16_tflops"] = round(3.0 * conv_tf, 1)
        roof["vs_fp32_mfma_peak"] = round(conv_tf / PEAK_FP32_MFMA_TF, 3)
        # informational (not `peak`): what the same three-product MFMA loop sustains on this part with nothing else running, on the convolutions'
        # operand statistics, under the 1400 W limit -- tools/micro/mfma_peak.hip, profiles/r06_mfma_sustained.txt (1954 TFLOP/s fp16 at 1.86 GHz)
        roof["sustained_mfma_only_tflops_fp32_equiv"] = 651.0
        roof["frac_of_sustained"] = round(conv_tf / 651.0, 4)
    if args.uniform:
        # no per-launch events were recorded: nothing to divide by -- the line carries the wall-clock figures only
        note = "--uniform: the timed region ran the product default only (no per-launch HIP events); run without the switch for the roofline objects"
        roof.update(achieved=None, frac=None, launches=0, avg_launch_ms=None, launches_by_abi={}, traffic=None, note=note)
        hbm, k5_k12, hbm_ms = {"note": note}, {"note": note, "kernel_ms": None}, None
    else:
        hbm, k5_k12 = grouping_roofline(tot, args.N, nrows, args.Ns, world, max(n_ev_steps, 1), row_split=row_split)
        hbm_ms = k5_k12["kernel_ms"]
    out = {
        "metric": "images/s embed + s/iter for NxN rerank+DBSCAN, N=16k, 1/2/4/8 GPU",
        "value": round(n_img / (ms_step * 1e-3), 2), "unit": "images/s (embedded + grouped per wall second, whole iteration)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": ("f32 (embed: split-half operands hi+lo on fp16 MFMA, fp32 accumulate, 6e-8 max error vs the fp32 reference features)" if split else
                                                     "f32 (embed, fp32 MFMA)") + " / exact int64 via int8 MFMA digits + f64 + f16 (distance, re-rank: half semantics, bit-exact)",
        "data": "synthetic: N(0,1) 256x128 images + seeded Kaiming ResNet-50 weights for the embed leg; clustered unit-norm 2048-d embeddings "
                "for the grouping leg (track '%s', tools/synth.py: %s) -- random-init backbone features are degenerate: reid/rerank.py:40 NaN path"
                % (args.track_g, "identity sizes 1..24, unequal spreads, 30 % confusable centres -> noise, border points, merged clusters"
                   if args.track_g == "hard" else "16 per identity, trivially separable (SURVEY.md 8d)"),
        "config": {"workload": "BASELINE configs[1]+[2]: N=%d target + Ns=%d source images -> ResNet-50 2048-d embed (orig+flip) -> "
                               "k-reciprocal re-rank (k1=20,k2=6,lambda=%.1f) -> eps rule (rho=%.1e) -> DBSCAN(min_samples=4), 1 feature split"
                               % (args.N, args.Ns, args.lambda_value, args.rho),
                   "N": args.N, "Ns": args.Ns, "d": 2048, "embed_batch": args.batch,
                   "parallelism": ("images sharded over %d GPU(s); grouping leg: %s" % (world, "NxN row blocks sharded, small tables all-gathered" if g_group is not None
                                   else ("the whole leg on every rank, no collective (dist.choose_grouping)" if world > 1 else "one GPU"))),
                   "grouping_form": grouping_form,
                   "timed_region": ("product default only (--uniform): no per-launch events" if args.uniform else
                                    "the K timed steps ALTERNATE two instrumentation modes: even steps bracket every grouping launch with HIP events and keep the "
                                    "re-rank on one stream, odd steps run the product default (source term on a second stream); in every step each 8th "
                                    "embedding batch runs its two forwards on one stream with per-launch events, the others on two streams -- ms_per_step "
                                    "is the blend (the modes differ by ~0.3 ms of ~1100); `--uniform` times the product default alone"),
                   "legs": "two synthetic tracks (SURVEY.md 8d): the embed leg embeds N(0,1) images; the grouping leg re-ranks resident clustered "
                           "embeddings of the same shape, NOT the embed leg's output (random-init features hit reid/rerank.py:40's NaN path); the "
                           "hand-off embed -> grouping is covered by tests/test_gpu_chain.py",
                   "embed_mode": "two HIP streams per batch (product default); every 8th batch on one stream for the per-launch events of `roofline`"},
        "embed_images_per_s": round(n_img / (t_embed * 1e-3), 1), "embed_ms": round(t_embed, 2),
        "rerank_dbscan_s_per_iter": round((t_rerank + t_cluster) * 1e-3, 5), "rerank_ms": round(t_rerank, 3), "eps_dbscan_ms": round(t_cluster, 3),
        "grouping_timing": grouping_timing,
        "labels": {"clusters": int(labels.max() + 1), "noise": int((labels < 0).sum()), "eps": eps,
                   "sha256": __import__("hashlib").sha256(np.ascontiguousarray(labels, dtype=np.int64).tobytes()).hexdigest()[:16]},
        "rank_mode": rerank.default_rank_mode(), "host_syncs": host_syncs, "collectives": collectives, "build": build_fingerprint(),
        "roofline": roof, "roofline_kernels": hbm,
        "roofline_k5_k12": k5_k12,
    }
    out.update(extras)
    if "rank_mode" in extras and "ssg_topk_rank_introsort" in tot and hbm_ms is not None:
        alt_ms = hbm_ms - tot["ssg_topk_rank_introsort"][1] / max(n_ev_steps, 1) + extras["rank_mode"]["stable_kernel_ms"]
        out["roofline_k5_k12"]["frac_with_stable_order"] = round(8.0 * nrows * args.N / (alt_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    if not args.no_cpu_baseline and world == 1:
        _lib._lib = timer.L
        out["cpu_baseline"] = cpu_baseline(args, emb_np[args.track_g][0], emb_np[args.track_g][1], labels, eps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
