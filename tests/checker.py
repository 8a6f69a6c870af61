"""What the GPU tests are checked against, reached through the checker process (oracle/sidecar.py).

``conftest.py`` starts that process before anything in pytest touches the GPU.  Everything here runs THERE:
the C oracle, the numpy layout restatement, the compiled reference, and every child command the tests need
(hipcc, python children) -- the process under test shares no heap with its checker and never forks.

``assert_same_rows`` is the comparison of the parity tests.  A mismatch is always a failure; the message
says which side is wrong, row by row, from the contract itself evaluated on the
read strings in plain Python -- independent of the HIP path AND of the oracle.
"""
from __future__ import annotations

import os
from collections import Counter
from typing import List, Optional, Sequence

import numpy as np

_sidecar = None


def start() -> None:
    global _sidecar
    if _sidecar is None:
        from oracle.sidecar import Sidecar
        _sidecar = Sidecar()


def stop() -> None:
    global _sidecar
    if _sidecar is not None:
        _sidecar.close()
        _sidecar = None


def sidecar():
    if _sidecar is None:  # (a test module run outside the normal session: late start, still a separate process)
        start()
    return _sidecar


def _plain(seqs: Sequence) -> List[bytes]:
    return [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]


def oracle_overlaps(seqs: Sequence, min_length: int) -> np.ndarray:
    """``oracle.overlap_oracle.oracle_overlaps`` in the checker process: sorted (n, 6) int64 rows."""
    return sidecar().call("oracle.overlap_oracle", "oracle_overlaps", _plain(seqs), int(min_length))


def reference_overlaps(seqs: Sequence, min_length: int, quiet: bool = False):
    return sidecar().call("oracle.overlap_oracle", "reference_overlaps", _plain(seqs), int(min_length), quiet)


def have_reference() -> bool:
    return sidecar().call("oracle.overlap_oracle", "have_reference")


def layout_vectorised(rows, lengths, **kwargs) -> dict:
    return sidecar().call("oracle.layout_oracle", "layout_vectorised", np.asarray(rows), np.asarray(lengths), **kwargs)


def layout_sequential(rows, lengths, **kwargs) -> dict:
    """``oracle.layout_oracle.layout_sequential`` in the checker process, plus ``edges_array`` = its edge dict as the
    sorted (n, 4) array the tests compare."""
    rows = [tuple(int(x) for x in r) for r in rows]
    want = sidecar().call("oracle.layout_oracle", "layout_sequential", rows, [int(x) for x in lengths], **kwargs)
    want["edges_array"] = sidecar().call("oracle.layout_oracle", "edges_dict_to_array", want["edges"])
    return want


def run(argv, **kwargs):
    """``subprocess.run`` in the checker process -> (returncode, stdout, stderr)."""
    return sidecar().run(list(argv), **kwargs)


def expected_multiplicity(seqs: Sequence[bytes], m: int, row) -> int:
    """How often the contract puts ``row`` into overlaps(m): 0, 1 or 2 (A and B family are not de-duplicated).
    Plain Python on the read strings (overlapper.cpp:64-116 as SURVEY.md section 8a-2 states it)."""
    a, b, s, e, bs, be = (int(x) for x in row)
    m = max(int(m), 1)
    if a == b or not (0 <= a < len(seqs)) or not (0 <= b < len(seqs)) or bs != 0:
        return 0
    A, B = seqs[a], seqs[b]
    la, lb = len(A), len(B)
    l = be
    if not (0 <= s < e <= la) or e - s != l or l < m or l > lb or A[s:e] != B[:l]:
        return 0
    n = 0
    if e == la:  # A family: the LONGEST suffix of a that is a prefix of b
        longest = next((k for k in range(min(la, lb), m - 1, -1) if A[la - k:] == B[:k]), None)
        n += 1 if longest == l else 0
    if l == lb:  # B family: every occurrence of the whole of b
        n += 1
    return n


def explain_difference(got: np.ndarray, want: np.ndarray, seqs: Optional[Sequence], m: int, limit: int = 12) -> str:
    cg, cw = Counter(map(tuple, got.tolist())), Counter(map(tuple, want.tolist()))
    diff = sorted(set((cg - cw).keys()) | set((cw - cg).keys()))
    lines = ["HIP %d rows, checker %d rows, %d distinct rows differ" % (len(got), len(want), len(diff))]
    hip_wrong = chk_wrong = 0
    plain = _plain(seqs) if seqs is not None else None
    for row in diff[:limit]:
        line = "  row %s: HIP x%d, checker x%d" % (row, cg[row], cw[row])
        if plain is not None:
            exp = expected_multiplicity(plain, m, row)
            hip_wrong += cg[row] != exp
            chk_wrong += cw[row] != exp
            line += ", contract x%d -> %s" % (exp, "HIP WRONG" if cg[row] != exp else "CHECKER WRONG")
        lines.append(line)
    if plain is not None:
        lines.append("verdict over the rows shown: HIP wrong on %d, checker wrong on %d" % (hip_wrong, chk_wrong))
    return "\n".join(lines)


def snapshot(seqs: Sequence) -> List[bytes]:
    """Real copies (separate memory) of the reads a test is about to hand to the library."""
    return [bytes(bytearray(s)) for s in _plain(seqs)]


def assert_inputs_unchanged(seqs, snap) -> None:
    """The reads a test hands to the library are immutable Python objects; the rows are checked against the SAME objects
    afterwards (checker process + the contract in plain Python).  If one of them reads differently after the call than
    before it, the comparison that follows would blame whichever side saw the other version -- say so instead, with the
    bytes that changed (seen once in round 3: the library's rows equalled the oracle's on the regenerated reads, while the
    test process's own copy of ONE read no longer did)."""
    for i, (s, c) in enumerate(zip(_plain(seqs), snap)):
        if s != c:
            sb, cb = bytes(s), bytes(c)
            where = [k for k in range(min(len(sb), len(cb))) if sb[k] != cb[k]]
            dump = os.environ.get("PHASM_MISMATCH_DIR")
            if dump:
                os.makedirs(dump, exist_ok=True)
                np.savez_compressed(os.path.join(dump, "input_changed_%d.npz" % os.getpid()), index=i,
                                    before=np.frombuffer(cb, dtype=np.uint8), after=np.frombuffer(sb, dtype=np.uint8))
            raise AssertionError("host memory of the test process changed under the call: read %d (%d bytes) differs from its "
                                 "copy taken before the call at %d byte offsets, first %s: before %r after %r"
                                 % (i, len(cb), len(where), where[:8], cb[where[0]:where[0] + 16] if where else b"",
                                    sb[where[0]:where[0] + 16] if where else b""))


def assert_same_rows(got: np.ndarray, want: np.ndarray, seqs: Optional[Sequence] = None, m: int = 1, ctx="") -> None:
    """Sorted-multiset equality of (n, 6) row arrays; a mismatch fails with a per-row verdict."""
    if got.shape == want.shape and np.array_equal(got, want):
        return
    dump = os.environ.get("PHASM_MISMATCH_DIR")
    if dump:
        os.makedirs(dump, exist_ok=True)
        np.savez_compressed(os.path.join(dump, "mismatch_%d.npz" % os.getpid()), got=got, want=want, m=m,
                            seqs=np.array(_plain(seqs), dtype=object) if seqs is not None else np.array([], dtype=object))
    raise AssertionError("%s\n%s" % (ctx, explain_difference(got, want, seqs, m)))


def oracle_overlaps_ex(seqs: Sequence, min_length: int, max_diff: int, band: int, anchor: int = 32) -> np.ndarray:
    """``oracle.extend_oracle.oracle_overlaps_ex`` (CPU restatement of po_overlaps_ex) in the checker process."""
    return sidecar().call("oracle.extend_oracle", "oracle_overlaps_ex", _plain(seqs), int(min_length), int(max_diff),
                          int(band), int(anchor))
