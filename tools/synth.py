"""Synthetic embeddings for the grouping leg (SURVEY.md 8d "Track G"), shared by tests/, bench.py and tools/.

clustered():       the survey's set -- P = N/16 identities of 16 samples, unit-norm centres + isotropic noise, renormalised.
                   Trivially separable: every identity is one DBSCAN cluster, no noise, tiny edge list.
hard_clustered():  what a real target set looks like to the grouping step: identity sizes from 1 to 24
                   (singletons and pairs can never reach min_samples=4 -> noise), per-identity spread varying by 2.7x,
                   30 % of the centres pulled towards another identity (confusable identities -> border points, merged
                   clusters, conflicts), samples shuffled so cluster numbering is not the generation order.
"""
import numpy as np


def clustered(N, d, seed, per_id=16, intra=0.5):
    rng = np.random.default_rng(seed)
    P = max(1, N // per_id)
    c = rng.standard_normal((P, d)); c /= np.linalg.norm(c, axis=1, keepdims=True)
    sigma = np.sqrt(intra / 2.0 / d)
    ids = np.arange(N) % P
    x = c[ids] + sigma * rng.standard_normal((N, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def hard_clustered(N, d, seed, intra=0.5, frac_single=0.15, sizes=(2, 3, 4, 8, 16, 24), return_ids=False):
    rng = np.random.default_rng(seed)
    ids, pid = [], 0
    n_single = int(N * frac_single)
    while len(ids) < N - n_single:
        s = int(rng.choice(sizes)); ids += [pid] * s; pid += 1
    ids = np.asarray(ids[:N - n_single] + list(range(pid, pid + n_single)))
    P = int(ids.max()) + 1
    c = rng.standard_normal((P, d)).astype(np.float32); c /= np.linalg.norm(c, axis=1, keepdims=True)
    mix = rng.random(P) < 0.3
    other = rng.integers(0, P, P)
    c[mix] = 0.8 * c[mix] + 0.6 * c[other[mix]]; c /= np.linalg.norm(c, axis=1, keepdims=True)
    sig = (np.sqrt(intra / 2.0 / d) * rng.uniform(0.6, 1.6, P)[ids])[:, None].astype(np.float32)
    x = c[ids] + sig * rng.standard_normal((N, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    perm = rng.permutation(N)
    x = np.ascontiguousarray(x[perm], dtype=np.float32)
    return (x, ids[perm]) if return_ids else x


def checkpoint_like_state_dict(seed):
    """Kaiming convolutions + BatchNorm statistics with the spread of a trained, folded checkpoint: per-channel scales
    gamma / sqrt(var + eps) log-normal over ~3 decades (rms 1 per layer so the network neither explodes nor dies),
    running_var log-uniform in [1e-3, 1e2], non-zero running_mean / beta.  Shared by the GPU tests and by tools/make_golden.py
    (tests/golden/embed_ckpt_ref.npz: the REAL reference model under these statistics)."""
    import torch
    import ssg_amd
    sd = ssg_amd.synthetic_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    for k in list(sd):
        if k.endswith("running_var") and k.startswith("base."):
            p = k[: -len("running_var")]
            n = sd[k].numel()
            var = torch.exp(torch.empty(n).uniform_(float(np.log(1e-3)), float(np.log(1e2)), generator=g))
            s = torch.exp(2.3 * torch.randn(n, generator=g))
            s = s / s.pow(2).mean().sqrt()
            sd[p + "running_var"] = var
            sd[p + "weight"] = s * torch.sqrt(var + 1e-5)
            sd[p + "running_mean"] = 0.1 * torch.randn(n, generator=g)
            sd[p + "bias"] = 0.05 * torch.randn(n, generator=g)
    return sd
