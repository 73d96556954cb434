"""
Fill-reducing orderings + symbolic LDL' analysis for the (fixed) KKT sparsity pattern of one
problem family.  Host side, runs once at code-generation time (the place where the reference
calls `osqp.OSQP().setup(...)`, `cvxpygen/solvers/osqp.py:126-131`, which runs AMD + QDLDL_etree).

The GPU triangular solves are level-scheduled, so besides fill the ordering is judged by the
height of the elimination tree (= number of dependent steps per solve).  `nested_dissection`
trades a little fill for a much shallower tree on chain-like (MPC) graphs.
"""

from __future__ import annotations

from typing import List, Tuple

import numpy as np
import scipy.sparse as sp


def _adjacency(K: sp.spmatrix) -> List[set]:
    K = sp.coo_matrix(K)
    n = K.shape[0]
    adj = [set() for _ in range(n)]
    for i, j in zip(K.row, K.col):
        if i != j:
            adj[i].add(int(j))
            adj[j].add(int(i))
    return adj


def min_degree(K: sp.spmatrix, nodes=None, adj=None) -> np.ndarray:
    """Plain minimum-degree on the explicit elimination graph (exact degrees, ties -> lowest
    index).  O(fill) set operations; fine for N up to a few thousand."""
    import heapq
    if adj is None:
        adj = _adjacency(K)
    n = len(adj)
    if nodes is None:
        nodes = range(n)
    nodes = list(nodes)
    inset = set(nodes)
    g = {v: set(u for u in adj[v] if u in inset) for v in nodes}
    heap = [(len(g[v]), v) for v in nodes]
    heapq.heapify(heap)
    done = set()
    order = []
    while heap:
        d, v = heapq.heappop(heap)
        if v in done or d != len(g[v]):
            continue
        done.add(v)
        order.append(v)
        nb = g.pop(v)
        for u in nb:
            g[u].discard(v)
        nbl = list(nb)
        for a in nbl:
            ga = g[a]
            before = len(ga)
            ga |= nb
            ga.discard(a)
            heapq.heappush(heap, (len(ga), a))
    return np.array(order, dtype=np.int64)


def _bfs_levels(adj, nodes_set, start):
    level = {start: 0}
    frontier = [start]
    order = [start]
    while frontier:
        nxt = []
        for v in frontier:
            for u in adj[v]:
                if u in nodes_set and u not in level:
                    level[u] = level[v] + 1
                    nxt.append(u)
                    order.append(u)
        frontier = nxt
    return level, order


def _components(adj, nodes_set):
    seen = set()
    comps = []
    for s in nodes_set:
        if s in seen:
            continue
        lvl, order = _bfs_levels(adj, nodes_set, s)
        seen.update(order)
        comps.append(order)
    return comps


def nested_dissection(K: sp.spmatrix, leaf_size: int = 48) -> np.ndarray:
    """Recursive bisection by BFS level structures from a pseudo-peripheral node; the middle
    level is the separator (ordered last).  Leaves are ordered by minimum degree."""
    adj = _adjacency(K)
    n = len(adj)

    def rec(nodes: List[int]) -> List[int]:
        if len(nodes) <= leaf_size:
            return list(min_degree(None, nodes=nodes, adj=adj))
        nodes_set = set(nodes)
        comps = _components(adj, nodes_set)
        if len(comps) > 1:
            out = []
            for c in sorted(comps, key=len):
                out += rec(c)
            return out
        # pseudo-peripheral start
        start = min(nodes, key=lambda v: len(adj[v]))
        for _ in range(3):
            lvl, order = _bfs_levels(adj, nodes_set, start)
            far = order[-1]
            if lvl[far] <= lvl.get(start, 0):
                break
            start = far
        lvl, order = _bfs_levels(adj, nodes_set, start)
        depth = max(lvl.values())
        if depth < 2:
            return list(min_degree(None, nodes=nodes, adj=adj))
        # choose the level closest to the median that gives a balanced split
        counts = np.bincount(list(lvl.values()), minlength=depth + 1)
        cum = np.cumsum(counts)
        half = len(nodes) / 2
        cand = [d for d in range(1, depth)]
        best = min(cand, key=lambda d: (abs(cum[d - 1] - (len(nodes) - cum[d])) + 4 * counts[d]))
        sep = [v for v in nodes if lvl[v] == best]
        left = [v for v in nodes if lvl[v] < best]
        right = [v for v in nodes if lvl[v] > best]
        if not left or not right:
            return list(min_degree(None, nodes=nodes, adj=adj))
        return rec(left) + rec(right) + list(min_degree(None, nodes=sep, adj=adj))

    perm = rec(list(range(n)))
    assert len(perm) == n and len(set(perm)) == n
    return np.array(perm, dtype=np.int64)


# ------------------------------------------------------------------------------------------------
def etree_and_counts(Ku: sp.csc_matrix) -> Tuple[np.ndarray, np.ndarray]:
    """QDLDL_etree restated (public algorithm: Liu's elimination-tree construction on the upper
    triangle, as QDLDL does): parent pointers and strict-lower column counts of L."""
    n = Ku.shape[0]
    Ap, Ai = Ku.indptr, Ku.indices
    work = np.full(n, -1, dtype=np.int64)
    etree = np.full(n, -1, dtype=np.int64)
    Lnz = np.zeros(n, dtype=np.int64)
    for j in range(n):
        work[j] = j
        for p in range(Ap[j], Ap[j + 1]):
            i = Ai[p]
            if i > j:
                raise ValueError('matrix must be upper triangular')
            while work[i] != j:
                if etree[i] == -1:
                    etree[i] = j
                Lnz[i] += 1
                work[i] = j
                i = etree[i]
    return etree, Lnz


def symbolic_ldl(Ku: sp.csc_matrix) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Structural pattern of L (strict lower, CSC, rows sorted) for K = L D L' without pivoting.
    Returns (Lp, Li, etree)."""
    n = Ku.shape[0]
    etree, Lnz = etree_and_counts(Ku)
    Lp = np.zeros(n + 1, dtype=np.int64)
    Lp[1:] = np.cumsum(Lnz)
    Li = np.zeros(Lp[-1], dtype=np.int64)
    fill = Lp[:-1].copy()
    Ap, Ai = Ku.indptr, Ku.indices
    mark = np.full(n, -1, dtype=np.int64)
    # row k of L = reach of the entries of column k of Ku in the etree (up-looking)
    for k in range(n):
        mark[k] = k
        for p in range(Ap[k], Ap[k + 1]):
            i = Ai[p]
            while mark[i] != k:
                mark[i] = k
                Li[fill[i]] = k          # L[k, i] != 0  -> row k appended to column i
                fill[i] += 1
                i = etree[i]
    assert np.all(fill == Lp[1:])
    # rows are appended in increasing k, so every column is already sorted
    return Lp, Li, etree


def etree_height(etree: np.ndarray) -> int:
    n = len(etree)
    depth = np.zeros(n, dtype=np.int64)
    for j in range(n - 1, -1, -1):
        p = etree[j]
        depth[j] = 0 if p < 0 else depth[p] + 1
    return int(depth.max()) + 1 if n else 0
