"""Adversarial keys for the introsort restatement (McIlroy, "A killer adversary for quicksort", 1999):
runs a sequential median-of-3 quicksort with the same pivot rule as numpy's aquicksort_ against a lazy
comparison oracle that fixes key values only when forced to, so that every partition is maximally
unbalanced.  The resulting key array drives numpy's argsort (and the device ranking kernel's walk)
past its depth limit into the heapsort fallback.  Test-input generator only."""
import numpy as np


def killer_keys(n):
    gas = n
    val = [gas] * n
    state = {"nsolid": 0, "cand": 0}

    def less(x, y):      # is key[x] < key[y] ?
        if val[x] == gas and val[y] == gas:
            if x == state["cand"]:
                val[x] = state["nsolid"]
            else:
                val[y] = state["nsolid"]
            state["nsolid"] += 1
        if val[x] == gas:
            state["cand"] = x
        elif val[y] == gas:
            state["cand"] = y
        return val[x] < val[y]

    a = list(range(n))
    stack = [(0, n - 1)]
    while stack:
        pl, pr = stack.pop()
        while pr - pl > 15:
            pm = pl + ((pr - pl) >> 1)
            if less(a[pm], a[pl]): a[pm], a[pl] = a[pl], a[pm]
            if less(a[pr], a[pm]): a[pr], a[pm] = a[pm], a[pr]
            if less(a[pm], a[pl]): a[pm], a[pl] = a[pl], a[pm]
            vp = a[pm]
            pi, pj = pl, pr - 1
            a[pm], a[pj] = a[pj], a[pm]
            while True:
                pi += 1
                while less(a[pi], vp): pi += 1
                pj -= 1
                while less(vp, a[pj]): pj -= 1
                if pi >= pj: break
                a[pi], a[pj] = a[pj], a[pi]
            a[pi], a[pr - 1] = a[pr - 1], a[pi]
            if pi - pl < pr - pi:
                stack.append((pi + 1, pr)); pr = pi - 1
            else:
                stack.append((pl, pi - 1)); pl = pi + 1
        for i in range(pl + 1, pr + 1):
            vi = a[i]; j = i
            while j > pl and less(vi, a[j - 1]):
                a[j] = a[j - 1]; j -= 1
            a[j] = vi
    for i in range(n):
        if val[i] == gas:
            val[i] = state["nsolid"]; state["nsolid"] += 1
    return np.asarray(val, dtype=np.uint16)
