"""TEST INFRASTRUCTURE ONLY -- CPU restatement of stage 1 of ``phasm layout``
(/root/reference/phasm/cli/assembler.py:52-139), the consumer of the overlap rows.

Two checkers:

* :func:`layout_sequential` -- the reference's control flow kept literally: one pass over the
  alignments IN ORDER through the stateful filter chain (``all(f(x) for f in filters)`` short-circuits,
  assembler.py:78-100), ``build_assembly_graph`` on the survivors (a dict of dicts stands in for
  networkx's adjacency; ``add_edge`` on an existing edge overwrites its attributes), then the removal
  of every filtered read in both orientations (assembler.py:108-126).  Pure Python loops.
* :func:`layout_vectorised` -- numpy, order-independent formulation (what the HIP kernels compute);
  used as ``bench.py``'s CPU baseline for this row and cross-checked against the sequential form.

PARITY PINNING.  The classification, overlap-length, overhang and filter decisions (and each filter's
``nodes_to_remove`` / ``filtered`` counter) are pinned by ``tests/golden/layout_cases.json``: outputs of
the reference's own classes (``phasm.alignments.LocalAlignment``, ``phasm.filter.*``,
``phasm.io.gfa.gfa2_line_to_la``) imported from /root/reference by
``tests/golden/make_layout_golden.py``.  The edge arithmetic of ``build_assembly_graph``
(assembly_graph.py:146-176, eight lines) is restated only: the reference function cannot run here
(it calls networkx 1.x's three-argument ``add_edge``; networkx 3.4 is installed) and the reference's
test for it needs ``tests/data/alignments.gfa``, which the repository does not ship.  For that part:
parity unpinned by execution, restated line by line below.

Nodes are integers: oriented read ``name+`` = 2*i, ``name-`` = 2*i+1 for segment i; ``reverse()`` = ^1.
Rows are (a, b, astart, aend, bstart, bend); ``lengths[node]`` is the read length.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

OVERLAP_AB, OVERLAP_BA, A_CONTAINED, B_CONTAINED = 0, 1, 2, 3   # AlignmentType, alignments.py:16-20


def classify(as_, ae, bs, be, la, lb) -> int:
    """LocalAlignment.classify, alignments.py:248-258."""
    if as_ <= bs and la - ae <= lb - be:
        return A_CONTAINED
    if as_ >= bs and la - ae >= lb - be:
        return B_CONTAINED
    if as_ >= bs:
        return OVERLAP_AB
    return OVERLAP_BA


def overlap_length(as_, ae, bs, be) -> int:
    """get_overlap_length, alignments.py:239-241."""
    return max(ae - as_, be - bs)


def overhang(as_, ae, bs, be, la, lb) -> int:
    """get_overhang, alignments.py:243-246."""
    return min(as_, bs) + min(la - ae, lb - be)


def edges_of(row, t, la, lb) -> List[Tuple[int, int, int, int]]:
    """The two add_edge calls of build_assembly_graph for one alignment, assembly_graph.py:146-176:
    (u, v, weight, overlap_len)."""
    a, b, as_, ae, bs, be = row
    ovl = overlap_length(as_, ae, bs, be)
    if t == OVERLAP_AB:
        return [(a, b, as_ - bs, ovl), (b ^ 1, a ^ 1, (lb - be) - (la - ae), ovl)]
    if t == OVERLAP_BA:
        return [(b, a, bs - as_, ovl), (a ^ 1, b ^ 1, (la - ae) - (lb - be), ovl)]
    return []


def layout_sequential(rows: Sequence[Sequence[int]], lengths: Sequence[int], min_read_length: int = 0,
                      min_overlap_length: int = 0, max_overhang_abs: int = 1000,
                      max_overhang_rel: float = 0.8) -> dict:
    """Literal restatement of assembler.py:78-126.  Returns
    ``{"types", "passed" (row indices that reached build_assembly_graph), "filters": [{name, filtered,
    nodes_to_remove}], "edges": {(u, v): (weight, overlap_len)}}``."""
    lengths = [int(x) for x in lengths]
    # filter chain in installation order (assembler.py:78-87); each: [name, filtered, nodes_to_remove]
    chain = [["ContainedReads", 0, set()]]
    if min_read_length:
        chain.append(["MinReadLength", 0, set()])
    if min_overlap_length:
        chain.append(["MinOverlapLength", 0, set()])
    chain.append(["MaxOverhang", 0, set()])

    def run(f, row, t, la, lb):
        name, _, removed = f
        a, b, as_, ae, bs, be = row
        if name == "ContainedReads":             # filter.py:90-101
            if t == A_CONTAINED:
                removed.add(a)
                return False
            if t == B_CONTAINED:
                removed.add(b)
                return False
            return not (a in removed or b in removed)
        if name == "MinReadLength":              # filter.py:45-58
            if la < min_read_length:
                removed.add(a)
                return False
            if lb < min_read_length:
                removed.add(b)
                return False
            return not (a in removed or b in removed)
        if name == "MinOverlapLength":           # filter.py:73-74
            return overlap_length(as_, ae, bs, be) >= min_overlap_length
        threshold = min(max_overhang_abs, max_overhang_rel * overlap_length(as_, ae, bs, be))  # filter.py:119-122
        return overhang(as_, ae, bs, be, la, lb) <= threshold

    types, passed = [], []
    adj: Dict[int, Dict[int, Tuple[int, int]]] = {}
    for i, row in enumerate(rows):
        row = tuple(int(x) for x in row)
        la, lb = lengths[row[0]], lengths[row[1]]
        t = classify(row[2], row[3], row[4], row[5], la, lb)
        types.append(t)
        ok = True
        for f in chain:                           # all(f(x) for f in filters): stops at the first False
            if not run(f, row, t, la, lb):
                f[1] += 1
                ok = False
                break
        if not ok:
            continue
        passed.append(i)
        for u, v, w, ovl in edges_of(row, t, la, lb):   # build_assembly_graph; add_edge overwrites
            adj.setdefault(u, {})
            adj.setdefault(v, {})
            adj[u][v] = (w, ovl)
    # assembler.py:108-126: every filtered read leaves the graph in both orientations
    for f in chain:
        for node in f[2]:
            for n in (node | 1, node & ~1):
                if n in adj:
                    del adj[n]
                    for u in adj:
                        adj[u].pop(n, None)
    edges = {(u, v): wv for u, nb in adj.items() for v, wv in nb.items()}
    return {"types": types, "passed": passed,
            "filters": [{"name": f[0], "filtered": f[1], "nodes_to_remove": sorted(f[2])} for f in chain],
            "edges": edges}


def layout_vectorised(rows: np.ndarray, lengths: np.ndarray, min_read_length: int = 0, min_overlap_length: int = 0,
                      max_overhang_abs: int = 1000, max_overhang_rel: float = 0.8) -> dict:
    """Order-independent numpy form: returns ``{"types", "contained" (bool per read name), "edges"
    ((n, 4) int64 array of (u, v, weight, overlap_len), sorted by (u, v))}``."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 6)
    lengths = np.asarray(lengths, dtype=np.int64)
    a, b, as_, ae, bs, be = rows.T
    la, lb = lengths[a], lengths[b]
    ra, rb = la - ae, lb - be
    t = np.full(len(rows), OVERLAP_BA, dtype=np.int64)
    t[as_ >= bs] = OVERLAP_AB
    t[(as_ >= bs) & (ra >= rb)] = B_CONTAINED
    t[(as_ <= bs) & (ra <= rb)] = A_CONTAINED
    contained = np.zeros(len(lengths) // 2, dtype=bool)
    contained[a[t == A_CONTAINED] >> 1] = True
    contained[b[t == B_CONTAINED] >> 1] = True
    ovl = np.maximum(ae - as_, be - bs)
    hang = np.minimum(as_, bs) + np.minimum(ra, rb)
    thr = np.minimum(float(max_overhang_abs), max_overhang_rel * ovl.astype(np.float64))
    ok = (t <= OVERLAP_BA) & (hang.astype(np.float64) <= thr)
    if min_read_length:
        ok &= (la >= min_read_length) & (lb >= min_read_length)
    if min_overlap_length:
        ok &= ovl >= min_overlap_length
    ok &= ~contained[a >> 1] & ~contained[b >> 1]
    idx = np.flatnonzero(ok)
    ab = t[idx] == OVERLAP_AB
    A, B = a[idx], b[idx]
    u1 = np.where(ab, A, B)
    v1 = np.where(ab, B, A)
    w1 = np.where(ab, as_[idx] - bs[idx], bs[idx] - as_[idx])
    u2 = np.where(ab, B ^ 1, A ^ 1)
    v2 = np.where(ab, A ^ 1, B ^ 1)
    w2 = np.where(ab, rb[idx] - ra[idx], ra[idx] - rb[idx])
    o = ovl[idx]
    # writer order = row order, edge 1 before edge 2; the last writer of a (u, v) owns it
    u = np.stack([u1, u2], 1).ravel()
    v = np.stack([v1, v2], 1).ravel()
    w = np.stack([w1, w2], 1).ravel()
    oo = np.stack([o, o], 1).ravel()
    key = (u << 32) | v
    order = np.argsort(key, kind="stable")
    ks = key[order]
    last = np.ones(len(ks), dtype=bool)
    last[:-1] = ks[1:] != ks[:-1]
    sel = order[last]
    edges = np.stack([u[sel], v[sel], w[sel], oo[sel]], 1) if len(sel) else np.empty((0, 4), dtype=np.int64)
    return {"types": t, "contained": contained, "edges": edges}


def edges_dict_to_array(edges: dict) -> np.ndarray:
    arr = np.array([(u, v, w, o) for (u, v), (w, o) in edges.items()], dtype=np.int64).reshape(-1, 4)
    return arr[np.lexsort((arr[:, 1], arr[:, 0]))] if len(arr) else arr
