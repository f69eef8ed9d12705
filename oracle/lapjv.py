"""ORACLE (test infrastructure): restatement of `lap.lapjv(cost, extend_cost=True, cost_limit=c)`.

`lap` (gatagat/lap, unpinned in the reference's requirements/runtime.txt:2) is an un-vendored third-party
dependency absent from /root/reference and from this image, so it is restated here from its published
algorithm [upstream-memory]: the dense Jonker-Volgenant solver of lap/_lapjv_cpp/lapjv.cpp (column reduction +
reduction transfer, two rounds of augmenting row reduction, shortest augmenting paths) and the cost-matrix
extension of lap/_lapjv.pyx.  Reference call sites: mmtrack/models/trackers/ocsort_tracker_disparity.py:260-261
and :312-313 (`cost, row, col = lap.lapjv(dists, extend_cost=True, cost_limit=1 - match_iou_thr)`).

Parity status: "parity unpinned" at this third-party boundary (no fixture of the reference pins it).  What the
tests do pin: the optimum (brute force over all assignments on <= 7x7 matrices) and, for matrices whose optimum
is NOT unique, WHICH optimal assignment this restatement returns (tests/golden/lapjv_ties.npz), which the product
solver must reproduce exactly.

Pure-Python loops on purpose (small cases only): every comparison below is the one the C++ source makes, in the
same order, on float64 — tie behaviour is a property of that order.
"""
import numpy as np

LARGE = 1000000.0


def _ccrrt_dense(n, cost, free_rows, x, y, v):
    """Column reduction and reduction transfer.  Returns the number of free rows."""
    for i in range(n):
        x[i] = -1
        v[i] = LARGE
        y[i] = 0
    for i in range(n):
        for j in range(n):
            c = cost[i][j]
            if c < v[j]:
                v[j] = c
                y[j] = i
    unique = [True] * n
    j = n
    while True:
        j -= 1
        i = y[j]
        if x[i] < 0:
            x[i] = j
        else:
            unique[i] = False
            y[j] = -1
        if not j > 0:
            break
    n_free_rows = 0
    for i in range(n):
        if x[i] < 0:
            free_rows[n_free_rows] = i
            n_free_rows += 1
        elif unique[i]:
            j = x[i]
            mn = LARGE
            for j2 in range(n):
                if j2 == j:
                    continue
                c = cost[i][j2] - v[j2]
                if c < mn:
                    mn = c
            v[j] -= mn
    return n_free_rows


def _carr_dense(n, cost, n_free_rows, free_rows, x, y, v):
    """Augmenting row reduction.  Returns the new number of free rows."""
    current = 0
    new_free_rows = 0
    rr_cnt = 0
    while current < n_free_rows:
        rr_cnt += 1
        free_i = free_rows[current]
        current += 1
        j1 = 0
        v1 = cost[free_i][0] - v[0]
        j2 = -1
        v2 = LARGE
        for j in range(1, n):
            c = cost[free_i][j] - v[j]
            if c < v2:
                if c >= v1:
                    v2 = c
                    j2 = j
                else:
                    v2 = v1
                    v1 = c
                    j2 = j1
                    j1 = j
        i0 = y[j1]
        v1_new = v[j1] - (v2 - v1)
        v1_lowers = v1_new < v[j1]
        if rr_cnt < current * n:
            if v1_lowers:
                v[j1] = v1_new
            elif i0 >= 0 and j2 >= 0:
                j1 = j2
                i0 = y[j2]
            if i0 >= 0:
                if v1_lowers:
                    current -= 1
                    free_rows[current] = i0
                else:
                    free_rows[new_free_rows] = i0
                    new_free_rows += 1
        else:
            if i0 >= 0:
                free_rows[new_free_rows] = i0
                new_free_rows += 1
        x[free_i] = j1
        y[j1] = free_i
    return new_free_rows


def _find_dense(n, lo, d, cols, y):
    """Find columns with minimum d[j] and put them on the SCAN list."""
    hi = lo + 1
    mind = d[cols[lo]]
    for k in range(hi, n):
        j = cols[k]
        if d[j] <= mind:
            if d[j] < mind:
                hi = lo
                mind = d[j]
            cols[k] = cols[hi]
            cols[hi] = j
            hi += 1
    return hi


def _scan_dense(n, cost, lo, hi, d, cols, pred, y, v):
    """Scan all columns in TODO starting from arbitrary column in SCAN and try to decrease d of the TODO columns
    using the SCAN column.  Returns (final_j or -1, lo, hi).  As in the C++ source, lo / hi are written back only
    when the scan list runs empty: on an early return (free column reached) the caller keeps the values it passed
    in, so `d[cols[lo]]` there still names a SCAN column."""
    lo_in, hi_in = lo, hi
    while lo != hi:
        j = cols[lo]
        lo += 1
        i = y[j]
        mind = d[j]
        h = cost[i][j] - v[j] - mind
        for k in range(hi, n):
            j = cols[k]
            cred_ij = cost[i][j] - v[j] - h
            if cred_ij < d[j]:
                d[j] = cred_ij
                pred[j] = i
                if cred_ij == mind:
                    if y[j] < 0:
                        return j, lo_in, hi_in
                    cols[k] = cols[hi]
                    cols[hi] = j
                    hi += 1
    return -1, lo, hi


def _find_path_dense(n, cost, start_i, y, v, pred):
    """Single iteration of the modified Dijkstra shortest path algorithm of the JV paper.  Returns the closest
    free column index."""
    lo = hi = 0
    final_j = -1
    n_ready = 0
    cols = list(range(n))
    d = [0.0] * n
    for i in range(n):
        pred[i] = start_i
        d[i] = cost[start_i][i] - v[i]
    while final_j == -1:
        if lo == hi:   # no columns left on the SCAN list
            n_ready = lo
            hi = _find_dense(n, lo, d, cols, y)
            for k in range(lo, hi):
                j = cols[k]
                if y[j] < 0:
                    final_j = j
        if final_j == -1:
            final_j, lo, hi = _scan_dense(n, cost, lo, hi, d, cols, pred, y, v)
    mind = d[cols[lo]]
    for k in range(n_ready):
        j = cols[k]
        v[j] += d[j] - mind
    return final_j


def _ca_dense(n, cost, n_free_rows, free_rows, x, y, v):
    """Augment along shortest paths from every remaining free row."""
    pred = [0] * n
    for f in range(n_free_rows):
        free_i = free_rows[f]
        i = -1
        j = _find_path_dense(n, cost, free_i, y, v, pred)
        while i != free_i:
            i = pred[j]
            y[j] = i
            j, x[i] = x[i], j
    return 0


def lapjv_internal(cost):
    """cost: (n, n) float64 -> x (row -> column), y (column -> row)."""
    n = len(cost)
    cost = [[float(c) for c in row] for row in cost]
    free_rows = [0] * n
    x, y, v = [0] * n, [0] * n, [0.0] * n
    ret = _ccrrt_dense(n, cost, free_rows, x, y, v)
    i = 0
    while ret > 0 and i < 2:
        ret = _carr_dense(n, cost, ret, free_rows, x, y, v)
        i += 1
    if ret > 0:
        ret = _ca_dense(n, cost, ret, free_rows, x, y, v)
    return x, y


def extend(cost, cost_limit):
    """The (n_rows + n_cols)^2 matrix lap builds for extend_cost=True with a finite cost_limit (_lapjv.pyx):
    original block top-left, cost_limit / 2 in the two off-diagonal blocks, 0 bottom-right."""
    cost = np.asarray(cost, dtype=np.float64)
    n_rows, n_cols = cost.shape
    n = n_rows + n_cols
    ext = np.empty((n, n), dtype=np.float64)
    ext[:] = cost_limit / 2.0
    ext[n_rows:, n_cols:] = 0
    ext[:n_rows, :n_cols] = cost
    return ext


def lapjv(cost, extend_cost=True, cost_limit=np.inf):
    """-> (opt, x, y) like lap.lapjv: x[i] = column assigned to row i or -1, y[j] = row of column j or -1
    (int32 arrays).  Only the extend_cost=True / finite cost_limit form the reference uses is restated.

    NaN costs: a NaN box (the empty-segment branch of extract_depth, ocsort_disparity.py:163-165) makes every
    comparison inside the C++ solver false; termination is then not guaranteed (undefined behaviour upstream).
    This restatement — and the product — make such entries unmatchable (cost 1e6 > any cost_limit) instead."""
    assert extend_cost and np.isfinite(cost_limit)
    cost = np.asarray(cost, dtype=np.float64)
    cost = np.where(np.isnan(cost), 1e6, cost)
    n_rows, n_cols = cost.shape
    x, y = lapjv_internal(extend(cost, cost_limit))
    x = np.asarray(x, dtype=np.int32)
    y = np.asarray(y, dtype=np.int32)
    x[x >= n_cols] = -1
    y[y >= n_rows] = -1
    x = x[:n_rows]
    y = y[:n_cols]
    opt = cost[np.nonzero(x != -1)[0], x[x != -1]].sum()
    return opt, x, y
