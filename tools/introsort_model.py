"""CPU model of the device ranking kernel's algorithm (csrc/topk_intro.hip).

numpy's default `argsort` (reid/rerank.py:70) is an unstable introsort on an index array
(npysort/quicksort.cpp aquicksort_<half>): which of several equal keys lands in column r of
initial_rank depends on the whole sequence of Hoare partitions.  The device kernel reproduces
that sequence exactly, but (a) only walks the sub-ranges that intersect the first K output
columns and (b) executes each partition as a data-parallel rank/pair computation instead of
the two sequential scanning pointers.  This file states (b) in numpy so that it can be checked
against the sequential restatement (oracle ora_argsort_half) on the CPU:

  after median-of-3 and the pivot swap, with S = [pl+1, pr-2]:
    left stoppers   L_1 < L_2 < ...   positions p in S with not key[p] <  vp
    right stoppers  R_1 > R_2 > ...   positions p in S with not key[p] >  vp
    m   = #{k : L_k < R_k}                      (number of swaps the scanning pointers perform)
    swap A[L_k] <-> A[R_k] for k <= m           (all pairs are disjoint)
    pi  = min(L_{m+1}, R_m)   with L_{last+1} := pr-1 (the parked pivot) and R_0 := pr-1
    swap A[pi] <-> A[pr-1]
  m is found without materialising the lists: with f(p) = #L-stoppers at positions <= p and
  g(p) = #R-stoppers at positions > p,  m = f(p*) for p* = the last position with g(p) >= f(p).
"""
import numpy as np

SMALL = 15   # ranges with pr - pl > 15 are partitioned (numpy 2.2.6, probed); else insertion sort


def _heapsort(key, a, lo, n):
    """numpy aheapsort_ on a[lo:lo+n] (1-based heap), restated."""
    b = a[lo:lo + n].copy()
    h = np.concatenate([[0], b])   # 1-based
    less = lambda x, y: key[x] < key[y]
    l = n >> 1
    while l > 0:
        tmp = h[l]; i = l; j = l << 1
        while j <= n:
            if j < n and less(h[j], h[j + 1]):
                j += 1
            if less(tmp, h[j]):
                h[i] = h[j]; i = j; j += j
            else:
                break
        h[i] = tmp
        l -= 1
    nn = n
    while nn > 1:
        tmp = h[nn]; h[nn] = h[1]; nn -= 1
        i = 1; j = 2
        while j <= nn:
            if j < nn and less(h[j], h[j + 1]):
                j += 1
            if less(tmp, h[j]):
                h[i] = h[j]; i = j; j += j
            else:
                break
        h[i] = tmp
    a[lo:lo + n] = h[1:]


def partition_parallel(key, a, pl, pr):
    """One Hoare partition of a[pl..pr] (index array a, keys key[a[.]]) in rank/pair form. Returns pi."""
    k = lambda p: key[a[p]]
    pm = pl + ((pr - pl) >> 1)
    if k(pm) < k(pl): a[pm], a[pl] = a[pl], a[pm]
    if k(pr) < k(pm): a[pr], a[pm] = a[pm], a[pr]
    if k(pm) < k(pl): a[pm], a[pl] = a[pl], a[pm]
    vp = k(pm)
    a[pm], a[pr - 1] = a[pr - 1], a[pm]
    s0, s1 = pl + 1, pr - 2            # scan region, inclusive
    ks = key[a[s0:s1 + 1]]
    isL = ks >= vp
    isR = ks <= vp
    f = np.cumsum(isL)                                  # L-stoppers at positions <= p
    g = isR[::-1].cumsum()[::-1] - isR                  # R-stoppers at positions > p
    ok = g >= f
    m = int(f[np.nonzero(ok)[0][-1]]) if ok.any() else 0
    Lpos = s0 + np.nonzero(isL)[0]
    Rpos = (s0 + np.nonzero(isR)[0])[::-1]
    if m:
        l, r = Lpos[:m], Rpos[:m]
        assert np.all(l < r)
        a[l], a[r] = a[r].copy(), a[l].copy()
    Lnext = int(Lpos[m]) if m < len(Lpos) else pr - 1
    Rm = int(Rpos[m - 1]) if m >= 1 else pr - 1
    pi = min(Lnext, Rm)
    a[pi], a[pr - 1] = a[pr - 1], a[pi]
    return pi


def argsort_topk(key, K):
    """First K entries of numpy's default argsort of `key` (1-D uint16 order keys, e.g. the
    bit patterns of non-negative halves), walking only the ranges that intersect [0, K)."""
    n = len(key)
    a = np.arange(n, dtype=np.int64)
    depth0 = 2 * (int(n).bit_length() - 1)
    stack = [(0, n - 1, depth0, True)]
    while stack:
        pl, pr, cd, chk = stack.pop()
        if chk and cd < 0:
            _heapsort(key, a, pl, pr - pl + 1)
            continue
        needed = True
        while pr - pl > SMALL:
            pi = partition_parallel(key, a, pl, pr)
            cd -= 1
            left, right = (pl, pi - 1), (pi + 1, pr)
            if pi - pl < pr - pi:
                cont, pushed = left, right
            else:
                cont, pushed = right, left
            if pushed[0] < K and pushed[0] <= pushed[1]:
                stack.append((pushed[0], pushed[1], cd, True))
            if cont[0] < K and cont[0] <= cont[1]:
                pl, pr = cont
            else:
                needed = False
                break
        if needed and pr > pl:
            seg = a[pl:pr + 1]
            order = np.argsort(key[seg], kind="stable")     # insertion sort == stable sort of the range
            a[pl:pr + 1] = seg[order]
    return a[:K]
