"""ORACLE (test infrastructure, never shipped): the assignment step of eval/evaluation.py:206-208, i.e. the solver the
reference vendors in eval/munkres.py (Kuhn-Munkres in Clapper's six steps), restated as scalar loops over Python lists.

The oracle owns this solver: oracle/metrics_np.py must not borrow the product's (quber_amd/eval/assignment.py, numpy
masks), otherwise the unpinned boundary half of the metrics would compare the product's assignment with itself.

Order of choices that decides among equally good assignments (it changes `obj_mIOU`), each as the reference does it:
  * step 1 reduces rows only (munkres.py:385-399);
  * step 2 stars greedily in row-major order (munkres.py:401-418);
  * the zero primed in step 4 is the one in the FIRST row that has an uncovered zero, at the LAST such column of that
    row - the scan of munkres.py:536-560 does not leave the row at its first hit;
  * step 6 adds to covered rows and subtracts from uncovered columns in one sweep (munkres.py:510-524);
  * the matrix is padded to square with zeros (munkres.py:271-318) and only pairs inside the original shape are
    returned, in row order (munkres.py:365-372).

Pinned: tests/golden/munkres_expected.json (assignments of the vendored solver, made by oracle/gen_golden.py)."""


def assign(cost):
    """cost: rows x cols nested sequence / array of non-negative numbers -> [(row, col)] of the optimal assignment."""
    rows = len(cost)
    cols = len(cost[0]) if rows else 0
    n = max(rows, cols)
    c = [[float(cost[i][j]) if (i < rows and j < cols) else 0.0 for j in range(n)] for i in range(n)]
    for row in c:
        lo = min(row)
        for j in range(n):
            row[j] -= lo
    STAR, PRIME = 1, 2
    mark = [[0] * n for _ in range(n)]
    row_cov, col_cov = [False] * n, [False] * n
    for i in range(n):
        for j in range(n):
            if c[i][j] == 0 and not row_cov[i] and not col_cov[j]:
                mark[i][j] = STAR
                row_cov[i] = col_cov[j] = True

    def first(seq, val):
        for k, v in enumerate(seq):
            if v == val:
                return k
        return -1

    while True:
        row_cov = [False] * n
        col_cov = [any(mark[i][j] == STAR for i in range(n)) for j in range(n)]
        if sum(col_cov) >= n:
            break
        # step 4 (with step 6 whenever no uncovered zero is left)
        while True:
            zi = zj = -1
            for i in range(n):
                if row_cov[i]:
                    continue
                for j in range(n):
                    if c[i][j] == 0 and not col_cov[j]:
                        zi, zj = i, j            # keeps overwriting: last such column of the row
                if zi >= 0:
                    break
            if zi < 0:
                m = min(c[i][j] for i in range(n) if not row_cov[i] for j in range(n) if not col_cov[j])
                for i in range(n):
                    for j in range(n):
                        if row_cov[i]:
                            c[i][j] += m
                        if not col_cov[j]:
                            c[i][j] -= m
                continue
            mark[zi][zj] = PRIME
            sj = first(mark[zi], STAR)
            if sj < 0:
                break
            row_cov[zi] = True
            col_cov[sj] = False
        # step 5: alternating path from the primed zero, then flip it
        path = [(zi, zj)]
        while True:
            si = first([mark[i][path[-1][1]] for i in range(n)], STAR)
            if si < 0:
                break
            path.append((si, path[-1][1]))
            path.append((si, first(mark[si], PRIME)))
        for i, j in path:
            mark[i][j] = 0 if mark[i][j] == STAR else STAR
        for i in range(n):
            for j in range(n):
                if mark[i][j] == PRIME:
                    mark[i][j] = 0
    return [(i, j) for i in range(rows) for j in range(cols) if mark[i][j] == STAR]
