"""TEST INFRASTRUCTURE ONLY -- Python face of oracle/extend_oracle.c, the CPU restatement of the banded
seed-extension mode (``po_overlaps_ex``).  PARITY UNPINNED for ``max_diff > 0`` (the reference is exact:
/root/reference/src/overlapper.cpp:28-150); with ``max_diff = 0`` it is the exact contract and is pinned by the
reference goldens (tests/test_oracle.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Sequence

import numpy as np

from .overlap_oracle import ROW_DTYPE, _as_bytes, sort_rows, struct_to_rows

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libextend_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "extend_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libextend_oracle.so"])
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_overlaps_ex.restype = ctypes.c_int
        lib.oracle_overlaps_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.oracle_ex_free.argtypes = [ctypes.c_void_p]
        lib.oracle_pair_hamming.restype = None
        lib.oracle_pair_hamming.argtypes = [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_void_p] * 4
        lib.oracle_extend_pairs.restype = None
        lib.oracle_extend_pairs.argtypes = [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        _lib = lib
    return _lib


def oracle_overlaps_ex(seqs: Sequence, min_length: int, max_diff: int, band: int, anchor: int = 32) -> np.ndarray:
    """Sorted (n, 6) rows of the extension mode.  ``anchor`` = the library's anchor length for the encoding
    (32 bases for 2-bit reads, 8 for 8-bit reads), capped at ``min_length`` like the library does."""
    lib = _load()
    seqs = [_as_bytes(s) for s in seqs]
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    offs = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    cat = np.frombuffer(b"".join(seqs) + b"\0" * 64, dtype=np.uint8)
    rows_p = ctypes.c_void_p()
    n = ctypes.c_uint64()
    rc = lib.oracle_overlaps_ex(cat.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(seqs), int(min_length),
                                int(max_diff), int(band), int(anchor), ctypes.byref(rows_p), ctypes.byref(n))
    if rc != 0:
        raise MemoryError("oracle_overlaps_ex failed")
    if n.value:
        buf = (ctypes.c_char * (n.value * ROW_DTYPE.itemsize)).from_address(rows_p.value)
        arr = np.frombuffer(buf, dtype=ROW_DTYPE).copy()
    else:
        arr = np.empty(0, dtype=ROW_DTYPE)
    lib.oracle_ex_free(rows_p)
    return sort_rows(struct_to_rows(arr))


def pair_hamming(cat: np.ndarray, ax: np.ndarray, ay: np.ndarray, n: np.ndarray) -> np.ndarray:
    """Hamming distance of ``cat[ax[k]:ax[k]+n[k]]`` and ``cat[ay[k]:ay[k]+n[k]]`` for every listed pair (all host cores)."""
    lib = _load()
    cat = np.ascontiguousarray(cat, dtype=np.uint8)
    ax = np.ascontiguousarray(ax, dtype=np.uint64)
    ay = np.ascontiguousarray(ay, dtype=np.uint64)
    n = np.ascontiguousarray(n, dtype=np.uint32)
    assert len(ax) == len(ay) == len(n)
    assert not len(n) or (int((ax + n).max()) <= len(cat) and int((ay + n).max()) <= len(cat))
    out = np.zeros(len(n), dtype=np.uint32)
    lib.oracle_pair_hamming(cat.ctypes.data, len(n), ax.ctypes.data, ay.ctypes.data, n.ctypes.data, out.ctypes.data)
    return out


def extend_pairs(cat: np.ndarray, ax: np.ndarray, rem: np.ndarray, ay: np.ndarray, lb: np.ndarray, max_diff: int, band: int) -> np.ndarray:
    """The restatement's DP (``extend_one``) on listed candidates: (n, 4) uint32 ``okA, jA, okB, iB`` -- x = ``cat[ax:ax+rem]``
    (a from the anchor position to its end), y = ``cat[ay:ay+lb]`` (all of b)."""
    lib = _load()
    cat = np.ascontiguousarray(cat, dtype=np.uint8)
    ax = np.ascontiguousarray(ax, dtype=np.uint64)
    ay = np.ascontiguousarray(ay, dtype=np.uint64)
    rem = np.ascontiguousarray(rem, dtype=np.uint32)
    lb = np.ascontiguousarray(lb, dtype=np.uint32)
    assert len(ax) == len(ay) == len(rem) == len(lb)
    assert not len(ax) or (int((ax + rem).max()) <= len(cat) and int((ay + lb).max()) <= len(cat))
    out = np.zeros((len(ax), 4), dtype=np.uint32)
    lib.oracle_extend_pairs(cat.ctypes.data, len(ax), ax.ctypes.data, rem.ctypes.data, ay.ctypes.data, lb.ctypes.data,
                            int(max_diff), int(band), out.ctypes.data)
    return out
