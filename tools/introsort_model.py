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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5: the OUT-OF-PLACE form of a partition for rows that do not fit in LDS (csrc/topk_intro.hip, stream_partition).
#
# The in-place replay above swaps entries of the row where it lies (a 512 KB global arena at N = 128 000: every swap is a random
# access).  On the main path of the walk only the LEFT child of a partition is ever needed (the ranges that intersect columns [0, K)
# all start at 0 until the pivot itself lands inside [0, K)), and the left child after the m swaps is
#       left[0]  = the smallest of the three median candidates,
#       left[p]  = A[p]                          for the positions p <= p* that are not L-stoppers,
#       left[p]  = A[R_k]   (k = rank of p)      for the k-th L-stopper from the left: the k-th R-stopper from the RIGHT,
# with A = the source array as the median-of-3 step left it (two overridden positions inside the part that is read: 0 and pm).
# So the source is only READ (streaming, coalesced), the left child is WRITTEN once (to LDS when it fits, else to a second global
# buffer), and the pairing goes through a rank-chunked staging buffer: for ranks [k0, k1] the R-stoppers are collected in rank order
# (pass X: mask words from the right end), then handed to the L-stoppers of the same ranks (pass Y: mask words from the left end).
# This function states that computation at the level of the kernel's 64-bit mask words, in the kernel's "q space" (bit q = p + qs:
# the source starts qs entries into its first 8-entry group), and is checked against partition_parallel above.

def _popc(x):
    return bin(x).count("1")


def stream_partition_left(key, a, n, qs=0, C=64):
    """left child [0, pi) of the Hoare partition of a[0:n] as the streamed kernel computes it -> (pi, left array)"""
    pr = n - 1
    pm = pr >> 1
    k = lambda e: key[e]
    el, em, er, e2 = int(a[0]), int(a[pm]), int(a[pr]), int(a[pr - 1])
    if k(em) < k(el): el, em = em, el
    if k(er) < k(em): er, em = em, er
    if k(em) < k(el): el, em = em, el
    vp = k(em)

    def A(p):                      # the source as the median-of-3 step leaves it, for the positions that are ever read (p <= pr - 2)
        return el if p == 0 else (e2 if p == pm else int(a[p]))
    s0, s1 = 1, pr - 2
    W = (n + qs + 63) >> 6
    L = [0] * W; R = [0] * W
    for p in range(s0, s1 + 1):
        q = p + qs
        x = k(A(p))
        if x >= vp: L[q >> 6] |= 1 << (q & 63)
        if x <= vp: R[q >> 6] |= 1 << (q & 63)
    cL = [_popc(w) for w in L]; cR = [_popc(w) for w in R]
    totL, totR = sum(cL), sum(cR)
    beforeL = [0] * W; beforeR = [0] * W            # stoppers in words < w
    for w in range(1, W):
        beforeL[w] = beforeL[w - 1] + cL[w - 1]; beforeR[w] = beforeR[w - 1] + cR[w - 1]
    # crossing word: number of leading words at whose END g >= f still holds
    wstar = 0
    while wstar < W and (totR - (beforeR[wstar] + cR[wstar])) >= (beforeL[wstar] + cL[wstar]):
        wstar += 1
    if wstar >= W:
        m, pstar = totL, s1
    else:
        best = -1
        for b in range(64):
            le = (2 << b) - 1
            f = beforeL[wstar] + _popc(L[wstar] & le)
            g = (totR - beforeR[wstar] - cR[wstar]) + _popc(R[wstar] & ~le & ((1 << 64) - 1))
            if g >= f:
                best = b
        if best < 0:
            m, pstar = beforeL[wstar], (wstar << 6) - 1 - qs
        else:
            m, pstar = beforeL[wstar] + _popc(L[wstar] & ((2 << best) - 1)), (wstar << 6) + best - qs
    pstar = max(s0 - 1, min(pstar, s1))
    pi = pstar + 1
    left = np.empty(pi, dtype=np.int64)
    for p in range(0, pi):                              # pass E1: everything, L-stopper positions are overwritten below
        left[p] = A(p)
    for k0 in range(1, m + 1, C):                       # rank chunks
        k1 = min(m, k0 + C - 1)
        buf = [None] * (k1 - k0 + 1)
        for w in range(W):                              # pass X (the kernel visits only the words that hold these ranks)
            for b in range(64):
                if (R[w] >> b) & 1:
                    leftrank = beforeR[w] + _popc(R[w] & ((2 << b) - 1))
                    kk = totR - leftrank + 1
                    if k0 <= kk <= k1:
                        buf[kk - k0] = A((w << 6) + b - qs)
        for w in range(W):                              # pass Y
            for b in range(64):
                if (L[w] >> b) & 1:
                    kk = beforeL[w] + _popc(L[w] & ((2 << b) - 1))
                    if k0 <= kk <= k1:
                        left[(w << 6) + b - qs] = buf[kk - k0]
    return pi, left


def _self_test_stream(trials=400, seed=7):
    rng = np.random.default_rng(seed)
    for t in range(trials):
        n = int(rng.integers(17, 700))
        kind = t % 5
        if kind == 0: key = rng.integers(0, 4, n)
        elif kind == 1: key = rng.integers(0, 1000, n)
        elif kind == 2: key = np.sort(rng.integers(0, 50, n))
        elif kind == 3: key = np.sort(rng.integers(0, 50, n))[::-1].copy()
        else: key = np.full(n, 3)
        key = key.astype(np.int64)
        a = rng.permutation(n).astype(np.int64)
        ref = a.copy()
        pi_ref = partition_parallel(key, ref, 0, n - 1)
        for qs in (0, 3, 7):
            pi, left = stream_partition_left(key, a.copy(), n, qs=qs, C=int(rng.integers(1, 90)))
            assert pi == pi_ref, (n, kind, qs, pi, pi_ref)
            assert np.array_equal(left, ref[:pi]), (n, kind, qs)
    return True


if __name__ == "__main__":
    print("stream_partition_left == partition_parallel (left child):", _self_test_stream())
