"""GFA2 line emitter for the overlap file -- the wire format between ``phasm overlap`` and
``phasm layout`` / ``phasm phase``.

Mirrors the two helpers the reference's overlap command uses
(/root/reference/phasm/io/gfa.py:230-231 ``gfa_line`` and :234-240 ``gfa_header``): a line is
its fields joined by tabs plus a newline; the header is ``H\\tVN:z:2.0`` (no ``TS`` tag on this
path).  ``write_edges`` is the bulk form for the 24-byte row array: byte-identical to writing
``gfa_line("E", "*", a, b, astart, aend, bstart, bend, "*")`` per row
(phasm/cli/assembler.py:46-48).
"""
from __future__ import annotations

from typing import BinaryIO, Optional, Sequence, TextIO, Union

import numpy as np


def gfa_line(*args) -> str:
    return "\t".join(map(str, args)) + "\n"


def gfa_header(version: str = "2.0", trace_spacing: Optional[int] = None) -> str:
    parts = ["H", "VN:z:{}".format(version)]
    if trace_spacing:
        parts.append("TS:i:{:d}".format(trace_spacing))
    return gfa_line(*parts)


def write_edges(out: Union[TextIO, BinaryIO], rows: np.ndarray, ids: Sequence[str], chunk: int = 1 << 18) -> int:
    """Write one ``E`` line per row of the structured row array; returns the line count."""
    ids_arr = np.asarray(ids, dtype=object)
    n = len(rows)
    binary = "b" in getattr(out, "mode", "") or isinstance(out, (bytes, bytearray))
    for lo in range(0, n, chunk):
        r = rows[lo:lo + chunk]
        a = ids_arr[r["a_idx"]]
        b = ids_arr[r["b_idx"]]
        lines = ["E\t*\t%s\t%s\t%d\t%d\t%d\t%d\t*\n" % t
                 for t in zip(a, b, r["astart"].tolist(), r["aend"].tolist(),
                              r["bstart"].tolist(), r["bend"].tolist())]
        blob = "".join(lines)
        out.write(blob.encode("utf-8") if binary else blob)
    return n
