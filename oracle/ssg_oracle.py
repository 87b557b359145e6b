"""ctypes front-end of the CPU oracle (oracle/ssg_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ssg_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.

Function names mirror the reference call surface so parity tests read like the
reference's own call sites:
  re_ranking(...)        reid/rerank.py:27-127
  eps_rule(...)          selftraining.py:289-293
  dbscan(...)            selftraining.py:295,306 (sklearn 1.7.2 DBSCAN precomputed)
  compute_dist / generate_selflabel   selftraining.py:255-313
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libssg_oracle.so")
_lib = None

_u16p = ctypes.POINTER(ctypes.c_uint16)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile oracle/ssg_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ssg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.ora_source_vec.restype = ctypes.c_uint16
        _lib.ora_eps_f64.restype = ctypes.c_double
        _lib.ora_eps_f16.restype = ctypes.c_uint16
        _lib.ora_pairwise_sum_f32.restype = ctypes.c_float
        _lib.ora_pairwise_sum_f64.restype = ctypes.c_double
        _lib.ora_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def num_threads():
    return int(lib().ora_num_threads())


def set_num_threads(n):
    lib().ora_set_num_threads(int(n))


def half_exp_table():
    """Correctly rounded half exp(x) for all 65536 half bit patterns (as half)."""
    out = np.empty(65536, np.uint16)
    lib().ora_half_exp_table(_p(out, _u16p))
    return out.view(np.float16)


def pairwise_sum(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float64:
        return float(lib().ora_pairwise_sum_f64(_p(a, _f64p), ctypes.c_int64(a.size)))
    a = _c(a, np.float32)
    return np.float32(lib().ora_pairwise_sum_f32(_p(a, _f32p), ctypes.c_int64(a.size)))


def argsort_half(v, small=None):
    """numpy default (introsort) argsort of a 1-D half array, restated.  small: partition threshold (ranges with pr - pl > small
    are partitioned); None = 15, the behaviour of the numpy 2.2.6 build the goldens were generated with (numpy's source says
    16).  The probe test in tests/test_oracle_golden.py tells which one the installed numpy follows."""
    v = _c(v, np.float16)
    out = np.empty(v.size, np.int64)
    if small is not None:
        lib().ora_set_small_threshold(ctypes.c_int(int(small)))
    try:
        lib().ora_argsort_half(_p(v.view(np.uint16), _u16p), ctypes.c_int64(v.size), _p(out, _i64p))
    finally:
        if small is not None:
            lib().ora_set_small_threshold(ctypes.c_int(15))
    return out


def source_vec(tgt, src):
    """rerank.py:35-40 -> (v_raw half[N], v half[N], max half)."""
    tgt = _c(tgt, np.float32); src = _c(src, np.float32)
    N, d = tgt.shape
    v_raw = np.empty(N, np.uint16); v = np.empty(N, np.uint16)
    mx = lib().ora_source_vec(_p(tgt, _f32p), _p(src, _f32p), N, src.shape[0], d, _p(v_raw, _u16p), _p(v, _u16p))
    return v_raw.view(np.float16), v.view(np.float16), np.uint16(mx).view(np.float16)


def euclid(tgt, memory_save=False):
    """rerank.py:33,61-62 (memory_save: the MemorySave=True branch :49-59) -> D half [N,N]."""
    tgt = _c(tgt, np.float32)
    N, d = tgt.shape
    D = np.empty((N, N), np.uint16)
    lib().ora_euclid2(_p(tgt, _f32p), N, d, 1 if memory_save else 0, _p(D, _u16p))
    return D.view(np.float16)


def re_ranking(input_feature_source, input_feature, k1=20, k2=6, lambda_value=0.2, MemorySave=False,
               Minibatch=2000, no_rerank=False, rank_mode="introsort", stages=False):
    """Restatement of reid/rerank.py:27-127 re_ranking.

    rank_mode: 'introsort' (default) = numpy's default unstable argsort, i.e. the untouched reference;
               'stable' = canonical (value, index) order (numpy kind='stable').
    Returns (euclidean_dist half[N,N], final_dist f64[N,N] | None); with stages=True a
    dict of every stage boundary is returned as third element.
    """
    src = _c(input_feature_source, np.float32); tgt = _c(input_feature, np.float32)
    N, d = tgt.shape
    if no_rerank:
        return euclid(tgt, MemorySave), None
    K = min(max(k1 + 1, k2), N)
    E = np.empty((N, N), np.uint16)
    v = np.empty(N, np.uint16)
    rank = np.empty((N, K), np.int32)
    final = np.empty((N, N), np.float64)
    V = Vq = J = Jp = None
    if stages:
        V = np.empty((N, N), np.uint16); Vq = np.empty((N, N), np.uint16)
        J = np.empty((N, N), np.uint16); Jp = np.empty((N, N), np.uint16)
    nan_path = lib().ora_re_ranking(
        _p(src, _f32p), _p(tgt, _f32p), src.shape[0], N, d, int(k1), int(k2), ctypes.c_double(lambda_value),
        1 if rank_mode == "introsort" else 0, 1 if MemorySave else 0, _p(E, _u16p), _p(v, _u16p), _p(rank, _i32p),
        _p(V, _u16p), _p(Vq, _u16p), _p(J, _u16p), _p(Jp, _u16p), _p(final, _f64p))
    if stages:
        st = dict(v=v.view(np.float16), rank=rank, V=V.view(np.float16),
                  V_qe=(Vq.view(np.float16) if k2 != 1 else V.view(np.float16)),
                  jaccard=J.view(np.float16), jaccard_scaled=Jp.view(np.float16), nan_path=bool(nan_path))
        return E.view(np.float16), final, st
    return E.view(np.float16), final


def eps_rule(dist, rho):
    """selftraining.py:289-293.  f64 matrix -> python float; half matrix -> np.float16."""
    dist = np.ascontiguousarray(dist)
    N = dist.shape[0]
    cnt = ctypes.c_int64(0); top = ctypes.c_int64(0)
    if dist.dtype == np.float16:
        e = lib().ora_eps_f16(_p(dist.view(np.uint16), _u16p), N, ctypes.c_double(rho), ctypes.byref(cnt), ctypes.byref(top))
        return np.uint16(e).view(np.float16), cnt.value, top.value
    dist = _c(dist, np.float64)
    e = lib().ora_eps_f64(_p(dist, _f64p), N, ctypes.c_double(rho), ctypes.byref(cnt), ctypes.byref(top))
    return float(e), cnt.value, top.value


def dbscan(dist, eps, min_samples=4):
    """sklearn 1.7.2 DBSCAN(eps, min_samples, metric='precomputed').fit_predict(dist)."""
    dist = _c(dist, np.float64)
    N = dist.shape[0]
    labels = np.empty(N, np.int64)
    lib().ora_dbscan(_p(dist, _f64p), N, ctypes.c_double(float(eps)), int(min_samples), _p(labels, _i64p))
    return labels


def compute_dist(source_features, target_features, lambda_value, no_rerank, num_split=2, rank_mode="introsort"):
    """selftraining.py:255-277 (numpy in).  Unlike the reference, the euclidean matrices
    are kept when no_rerank=True so that the no-rerank path is runnable."""
    e_list, r_list = [], []
    if not isinstance(source_features, list):
        source_features, target_features = [source_features], [target_features]
    for s, t in zip(source_features, target_features):
        e, r = re_ranking(np.asarray(s), np.asarray(t), lambda_value=lambda_value, no_rerank=no_rerank, rank_mode=rank_mode)
        r_list.append(r)
        e_list.append(e if no_rerank else [])
    return e_list, r_list


def generate_selflabel(e_dist, r_dist, n_iter, rho, no_rerank, eps_list=None):
    """selftraining.py:280-313: eps rule at iteration 0 (frozen afterwards) + DBSCAN."""
    eps_list = [] if eps_list is None else eps_list
    labels_list = []
    for s in range(len(r_dist)):
        tmp = e_dist[s] if no_rerank else r_dist[s]
        if n_iter == 0:
            eps, _, _ = eps_rule(tmp, rho)
            eps_list.append(eps)
        labels_list.append(dbscan(tmp, eps_list[s], 4))
    return labels_list, eps_list


def re_ranking_init(query_feature, gallery_feature, k1=20, k2=6, lambda_value=0.3):
    """reid/rerank.py:171-234 (float32 cosine variant).  The dot products are taken with numpy
    (float32 BLAS, like the reference); everything after that is the C restatement."""
    q = _c(query_feature, np.float32); g = _c(gallery_feature, np.float32)
    qg = np.dot(q, g.T); qq = np.dot(q, q.T); gg = np.dot(g, g.T)
    return re_ranking_init_dist(qg, qq, gg, k1, k2, lambda_value)


def re_ranking_init_dist(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    """reid/rerank_initial.py:40-99 (takes the three dot-product matrices)."""
    dots = _c(np.concatenate([np.concatenate([q_q_dist, q_g_dist], axis=1), np.concatenate([q_g_dist.T, g_g_dist], axis=1)], axis=0), np.float32)
    N, nq = dots.shape[0], q_g_dist.shape[0]
    out = np.empty((nq, N - nq), np.float32)
    lib().ora_re_ranking_init(_p(dots, _f32p), N, nq, int(k1), int(k2), ctypes.c_float(lambda_value), _p(out, _f32p))
    return out


def re_ranking_plain(input_feature_source, input_feature, k=20, lambda_value=0.1, MemorySave=False, Minibatch=2000, stages=False):
    """Restatement of reid/rerank_plain.py:125-178 re_ranking (kNN-set Jaccard variant): source term and half
    original distance exactly as rerank.py (same lines), knn_bool[i] = {j != i : D[i,j] <= k-th smallest of row i}
    (:165-170, np.partition), jaccard = scipy cdist(bool, bool, 'jaccard') = |A xor B| / |A or B| in float64 (0 when
    both sets are empty) -> half (:173), final = f64(half(J * half(1-lambda))) + f64(half(v_i + v_k)) * lambda (:175).
    Returns (final_dist, final_dist) like the reference; stages=True adds a dict(knn, jaccard, v, euclid)."""
    src = _c(input_feature_source, np.float32); tgt = _c(input_feature, np.float32)
    _, v, _ = source_vec(tgt, src)                       # half [N], already divided by its max
    D = euclid(tgt)                                      # half [N, N]
    N = D.shape[0]
    thr = np.partition(D, k - 1, axis=1)[:, k - 1]
    knn = D <= thr[:, None]
    knn[np.arange(N), np.arange(N)] = False
    A = knn.astype(np.int32)
    c = A @ A.T                                          # |A_i and A_k|
    n = A.sum(axis=1)
    denom = n[:, None] + n[None, :] - c
    num = denom - c
    J = np.zeros((N, N), np.float64)
    np.divide(num, denom, out=J, where=denom != 0)
    J16 = J.astype(np.float16)
    source_dist = (v[None, :] + v[:, None]).astype(np.float64)      # half + half -> half, widened
    final = (J16 * (1 - lambda_value)) + source_dist * lambda_value   # half * python float stays half (NEP 50), then + float64
    if stages:
        return final, final, dict(knn=knn, jaccard=J16, v=v, euclid=D)
    return final, final
