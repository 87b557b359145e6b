"""Host-side mirror of the grouping helpers of selftraining.py (reference :255-313).

`compute_dist` and `generate_selflabel` keep the reference's names, argument order and
return structure so that selftraining.py can import them instead of its own definitions
(INTEGRATION.md).  Differences, all deliberate and documented in DESIGN.md:
  * distances stay on the GPU as `DistHandle`s unless `materialize=True`;
  * with no_rerank=True the euclidean matrices are kept (the reference discards them at
    selftraining.py:260-266 and then crashes in generate_selflabel on `[]`);
  * the cached "cluster" objects are `ssg_amd.cluster.DBSCAN` instances (eps frozen after
    iteration 0 exactly like selftraining.py:283-298).
"""
import numpy as np
import torch

from .cluster import DBSCAN, as_handle, eps_rule, eps_rule_dbscan  # noqa: F401
from .rerank import DeviceBackedArray, re_ranking_device


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def compute_dist(source_features, target_features, lambda_value, no_rerank, num_split=2, materialize=False, group=None, grouping="auto"):
    """selftraining.py:255-277.  Features: torch tensors (CPU or CUDA) or numpy arrays,
    a list of S+1 per-split tensors or a single tensor.  Returns (euclidean_dist_list,
    rerank_dist_list) with one entry per split.

    group (torch.distributed) + grouping: 'shard' = every rank computes its `shard_bounds` row block and the small tables are
    all-gathered; 'replicate' = every rank runs the whole problem on its own GPU with no collective at all (the features are
    replicated on every rank anyway) -- identical handles, identical labels on every rank; 'auto' (default) = `dist.choose_grouping`:
    the form its time model predicts to be faster for this N and world size (small problems replicate)."""
    euclidean_dist_list, rerank_dist_list = [], []
    if not isinstance(source_features, list):
        source_features, target_features = [source_features], [target_features]
    dev = _dev()
    for s, t in zip(source_features, target_features):
        s = torch.as_tensor(s).to(dev, torch.float32); t = torch.as_tensor(t).to(dev, torch.float32)
        row0, nrows, grp = 0, None, None
        if group is not None:
            import torch.distributed as dist
            from .dist import choose_grouping, shard_bounds
            if choose_grouping(t.shape[0], dist.get_world_size(group), grouping) == "shard":
                row0, row1 = shard_bounds(t.shape[0], dist.get_rank(group), dist.get_world_size(group))   # ragged N allowed
                nrows, grp = row1 - row0, group
        h = re_ranking_device(s, t, lambda_value=lambda_value, no_rerank=no_rerank, keep_euclid=no_rerank, row0=row0, nrows=nrows, group=grp,
                              validate=materialize)     # fused path: the status words are read by generate_selflabel's first round trip
        if materialize:
            from . import hostio
            if no_rerank:
                euclidean_dist_list.append(hostio.to_host(h.euclid).numpy()); rerank_dist_list.append(None)
            else:
                f = DeviceBackedArray.attach(hostio.final_dist_to_host(h).numpy(), h)
                euclidean_dist_list.append([]); rerank_dist_list.append(f)
        else:
            euclidean_dist_list.append(h if no_rerank else [])
            rerank_dist_list.append(None if no_rerank else h)
    return euclidean_dist_list, rerank_dist_list


def generate_selflabel(e_dist, r_dist, n_iter, args, cluster_list=[]):   # noqa: B006 (mutable default kept: reference :280)
    """selftraining.py:280-313: eps rule at iteration 0 (estimator cached, eps frozen), then
    `cluster.fit_predict` per split.  `args` needs `.no_rerank` and `.rho`."""
    labels_list = []
    for s in range(len(r_dist)):
        tmp_dist = e_dist[s] if args.no_rerank else r_dist[s]
        if n_iter == 0:
            # eps rule + first fit as ONE device chain with one read-back (cluster.eps_rule_dbscan: the same eps, labels and core samples
            # as eps_rule followed by DBSCAN.fit); the estimator is cached with its eps exactly like selftraining.py:283-298
            eps, _, _, labels, core = eps_rule_dbscan(tmp_dist, args.rho, min_samples=4)
            print('eps in cluster: {:.3f}'.format(eps))
            cluster = DBSCAN(eps=eps, min_samples=4, metric='precomputed', n_jobs=8)
            cluster.labels_, cluster.core_sample_indices_, cluster.n_features_in_ = labels, core, len(labels)
            cluster_list.append(cluster)
            print('Clustering and labeling...')
        else:
            cluster = cluster_list[s]
            print('Clustering and labeling...')
            labels = cluster.fit_predict(tmp_dist)
        num_ids = len(set(labels.tolist())) - 1
        print('Iteration {} have {} training ids'.format(n_iter + 1, num_ids))
        labels_list.append(labels)
    return labels_list, cluster_list


def select_labeled(labels_list):
    """selftraining.py:315-324 join: keep sample i iff no split labelled it -1.
    Returns (kept indices, [per-split label lists])."""
    lab = np.stack([np.asarray(l) for l in labels_list], axis=1) if len(labels_list) else np.zeros((0, 0), np.int64)
    keep = np.nonzero((lab != -1).all(axis=1))[0]
    return keep, lab[keep]


def generate_dataset(trainval, labels_list, iter_n=None):
    """The dataset half of selftraining.py:315-324 generate_dataloader: [(fname, [label of split 0, ...], 0)] for every image of
    `trainval` ([(fname, pid, cam)], loader order) that no split labelled -1; prints the reference's progress line when
    iter_n is given.  (The DataLoader / RandomIdentitySampler the reference wraps around it belong to the fine-tune phase.)"""
    keep, lab = select_labeled(labels_list)
    new_dataset = [(trainval[int(i)][0], [lab[r, s] for s in range(lab.shape[1])], 0) for r, i in enumerate(keep)]
    if iter_n is not None:
        print('Iteration {} have {} training images'.format(iter_n + 1, len(new_dataset)))
    return new_dataset
