"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle.

Three checkers, from slowest/most literal to fastest:

* :func:`brute_force`  -- the output contract of ``ExactOverlapper::overlaps``
  (/root/reference/src/overlapper.cpp:28-150) written as plain
  Python loops over byte strings.  Small cases only.
* :func:`oracle_overlaps` -- ``oracle/overlap_oracle.c`` through ctypes (hash + memcmp).
* :func:`reference_overlaps` -- the *reference itself*: ``oracle/_ref/ref_overlapper``
  (compiled by ``make -C oracle ref`` from /root/reference, never copied).  Available only
  where that binary exists.

PARITY PINNING: ``brute_force`` and ``oracle_overlaps`` are pinned by the committed
fixtures in ``tests/golden/`` (outputs of ``reference_overlaps``; generator script
``tests/golden/make_golden.py``) -- see ``tests/test_oracle.py``.

All functions return an ``(n, 6)`` int64 array of rows
``(a_idx, b_idx, astart, aend, bstart, bend)`` sorted lexicographically (a sorted multiset:
the reference's row order is implementation-defined, overlapper.cpp:30,:68).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile
import time
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboverlap_oracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "ref_overlapper")

ROW_DTYPE = np.dtype([("a_idx", "<u4"), ("b_idx", "<u4"), ("astart", "<i4"),
                      ("aend", "<i4"), ("bstart", "<i4"), ("bend", "<i4")])


def _as_bytes(s) -> bytes:
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


def sort_rows(rows: np.ndarray) -> np.ndarray:
    """Canonical sorted-multiset form: (n,6) int64, lexicographically sorted."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 6)
    if len(rows) == 0:
        return rows
    order = np.lexsort(rows.T[::-1])
    return rows[order]


def struct_to_rows(arr: np.ndarray) -> np.ndarray:
    out = np.empty((len(arr), 6), dtype=np.int64)
    for j, name in enumerate(ROW_DTYPE.names):
        out[:, j] = arr[name]
    return out


def brute_force(seqs: Sequence, min_length: int) -> np.ndarray:
    """Literal statement of the contract.  O(n^2 * L^2); tiny inputs only."""
    seqs = [_as_bytes(s) for s in seqs]
    m = max(int(min_length), 1)  # a suffix array has no empty suffix: m=0 acts as m=1
    rows: List[Tuple[int, ...]] = []
    for ai, a in enumerate(seqs):
        la = len(a)
        for bi, b in enumerate(seqs):
            if ai == bi:
                continue
            lb = len(b)
            # A: longest suffix of a that is a prefix of b (overlapper.cpp:64-91)
            for l in range(min(la, lb), m - 1, -1):
                if a[la - l:] == b[:l]:
                    rows.append((ai, bi, la - l, la, 0, l))
                    break
            # B: every occurrence of the whole of b in a (overlapper.cpp:95-115)
            if lb >= m:
                p = a.find(b)
                while p != -1:
                    rows.append((ai, bi, p, p + lb, 0, lb))
                    p = a.find(b, p + 1)
    return sort_rows(np.array(rows, dtype=np.int64).reshape(-1, 6))


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboverlap_oracle.so"])
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_overlaps_cat.restype = ctypes.c_int
        lib.oracle_overlaps_cat.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
            ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.oracle_free.argtypes = [ctypes.c_void_p]
        _lib = lib
    return _lib


def oracle_overlaps_struct(seqs: Sequence, min_length: int) -> Tuple[np.ndarray, float]:
    """Run the C restatement; returns (row struct array in emission order, seconds)."""
    lib = _load()
    seqs = [_as_bytes(s) for s in seqs]
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    offs = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    cat = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
    rows_p = ctypes.c_void_p()
    n = ctypes.c_uint64()
    t0 = time.perf_counter()
    rc = lib.oracle_overlaps_cat(cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                 len(seqs), int(min_length), ctypes.byref(rows_p), ctypes.byref(n))
    dt = time.perf_counter() - t0
    if rc != 0:
        raise MemoryError("oracle_overlaps_cat failed")
    if n.value:
        buf = (ctypes.c_char * (n.value * ROW_DTYPE.itemsize)).from_address(rows_p.value)
        arr = np.frombuffer(buf, dtype=ROW_DTYPE).copy()
    else:
        arr = np.empty(0, dtype=ROW_DTYPE)
    lib.oracle_free(rows_p)
    return arr, dt


def oracle_overlaps(seqs: Sequence, min_length: int) -> np.ndarray:
    arr, _ = oracle_overlaps_struct(seqs, min_length)
    return sort_rows(struct_to_rows(arr))


def have_reference() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def reference_overlaps(seqs: Sequence, min_length: int, quiet: bool = False):
    """Run the compiled *reference* on the reads.  Returns (sorted rows, seconds, nrows).

    Read ids handed to the reference are the decimal read indices, so rows come back as
    indices.  Sequences must not contain whitespace/newlines (the driver is line based).
    """
    if not have_reference():
        raise FileNotFoundError(REF_BIN)
    seqs = [_as_bytes(s) for s in seqs]
    with tempfile.NamedTemporaryFile("wb", suffix=".tsv", delete=False) as f:
        for i, s in enumerate(seqs):
            f.write(b"%d\t%s\n" % (i, s))
        path = f.name
    try:
        cmd = [REF_BIN, str(int(min_length)), path] + (["--quiet"] if quiet else [])
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    finally:
        os.unlink(path)
    info = dict(kv.split("=") for kv in p.stderr.decode().split() if "=" in kv)
    secs = float(info["ref_seconds"])
    nrows = int(info["ref_rows"])
    if quiet:
        return None, secs, nrows
    txt = p.stdout.decode()
    if txt.strip():
        rows = np.array([[int(x) for x in ln.split("\t")] for ln in txt.splitlines()], dtype=np.int64)
    else:
        rows = np.empty((0, 6), dtype=np.int64)
    return sort_rows(rows), secs, nrows
