"""Munkres (Hungarian) assignment with the exact tie-breaking of the solver the reference vendors in
eval/munkres.py (Clapper's 6-step formulation, used by eval/evaluation.py:206-208).  Restated on numpy masks:
row-minimum reduction only, greedy row-major starring, "first row with an uncovered zero, LAST such column in it"
when priming (munkres.py:544-568 keeps scanning the row after a hit), add-then-subtract update of step 6.
The choice among equally good assignments decides which zero-overlap pairs enter `obj_union[idx]`, so the order
matters for `obj_mIOU`; a different optimal solver (e.g. scipy) would not reproduce the reference's numbers."""
import numpy as np


def munkres_assign(cost):
    cost = np.asarray(cost, dtype=np.float64)
    rows, cols = cost.shape
    n = max(rows, cols)
    C = np.zeros((n, n), np.float64)
    C[:rows, :cols] = cost                                   # zero padding (munkres.py pad_matrix)
    C -= C.min(axis=1, keepdims=True)                        # step 1
    star = np.zeros((n, n), bool)
    prime = np.zeros((n, n), bool)
    rcov, ccov = np.zeros(n, bool), np.zeros(n, bool)
    for i in range(n):                                       # step 2
        for j in range(n):
            if C[i, j] == 0 and not rcov[i] and not ccov[j]:
                star[i, j] = True
                rcov[i] = ccov[j] = True
    while True:
        rcov[:] = False                                      # step 3
        ccov = star.any(axis=0)
        if star.sum() >= n:
            break
        while True:                                          # steps 4 / 6
            free = (C == 0) & ~rcov[:, None] & ~ccov[None, :]
            cand = np.flatnonzero(free.any(axis=1))
            if cand.size == 0:
                m = C[np.ix_(~rcov, ~ccov)].min()
                C[rcov, :] += m
                C[:, ~ccov] -= m
                continue
            i = cand[0]
            j = np.flatnonzero(free[i])[-1]
            prime[i, j] = True
            s = np.flatnonzero(star[i])
            if s.size == 0:
                break
            rcov[i] = True
            ccov[s[0]] = False
        path = [(i, j)]                                      # step 5: alternate starred / primed zeros
        while True:
            r = np.flatnonzero(star[:, path[-1][1]])
            if r.size == 0:
                break
            path.append((r[0], path[-1][1]))
            path.append((r[0], np.flatnonzero(prime[r[0]])[0]))
        for r, c in path:
            star[r, c] = not star[r, c]
        prime[:] = False
    return [(i, j) for i in range(rows) for j in range(cols) if star[i, j]]
