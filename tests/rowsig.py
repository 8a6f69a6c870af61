"""Order-independent comparison of row multisets too large to sort comfortably (tens of millions of rows).

``signature`` = (count, two independent 64-bit sums of per-row hashes): equal multisets give equal signatures, and a
multiset that differs in any row changes both sums (up to a 2^-128 accident).  ``explain`` names the rows that differ.
"""
import numpy as np


def _mix(x: np.ndarray, k: int) -> np.ndarray:
    x = (x ^ (x >> np.uint64(31))) * np.uint64(k)
    return x ^ (x >> np.uint64(29))


def row_hashes(rows: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        r = rows.astype(np.uint64)
        h = np.zeros(len(r), dtype=np.uint64)
        for j in range(6):
            h = _mix(h * np.uint64(0x9E3779B97F4A7C15) + r[:, j] + np.uint64(j + 1), 0xBF58476D1CE4E5B9)
        return h


def signature(rows: np.ndarray):
    """(count, sum of row hashes, sum of re-mixed row hashes), all mod 2^64."""
    with np.errstate(over="ignore"):
        h = row_hashes(rows)
        h2 = _mix(h, 0x94D049BB133111EB)
        return len(rows), int(h.sum(dtype=np.uint64)), int(h2.sum(dtype=np.uint64))


def add_signatures(*sigs):
    n = sum(s[0] for s in sigs)
    return n, sum(s[1] for s in sigs) % (1 << 64), sum(s[2] for s in sigs) % (1 << 64)


def explain(got: np.ndarray, want: np.ndarray, limit: int = 8) -> str:
    """Which rows (with multiplicity) are missing from / extra in ``got``; for the failure message only."""
    hg, hw = row_hashes(got), row_hashes(want)
    ug, cg = np.unique(hg, return_counts=True)
    uw, cw = np.unique(hw, return_counts=True)
    allh = np.union1d(ug, uw)
    ng = np.zeros(len(allh), dtype=np.int64)
    nw = np.zeros(len(allh), dtype=np.int64)
    ng[np.searchsorted(allh, ug)] = cg
    nw[np.searchsorted(allh, uw)] = cw
    missing = allh[nw > ng]
    extra = allh[ng > nw]
    out = ["%d rows got, %d wanted; %d distinct rows missing, %d distinct rows extra" % (len(got), len(want), len(missing), len(extra))]
    for name, hs, src, hsrc in (("missing", missing, want, hw), ("extra", extra, got, hg)):
        for h in hs[:limit].tolist():
            out.append("  %s %s" % (name, src[np.nonzero(hsrc == np.uint64(h))[0][0]].tolist()))
    return "\n".join(out)


def assert_same_multiset(got: np.ndarray, want: np.ndarray, what: str = "") -> None:
    if signature(got) != signature(want):
        raise AssertionError("%s: row multisets differ\n%s" % (what, explain(got, want)))
