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
