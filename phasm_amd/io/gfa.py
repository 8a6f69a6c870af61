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


# ---- reading side: what `phasm layout` takes from the overlap file ----------------------------

def _gfa_pos_to_int(pos: str) -> int:
    """/root/reference/phasm/io/gfa.py:65-69: a position may carry the GFA2 end marker ``$``."""
    return int(pos[:-1]) if pos.endswith("$") else int(pos)


def gfa2_parse_segment(line: str):
    """``S <id> <length> <sequence|*>`` -> (id, length); mirrors gfa2_segment_to_read (gfa.py:33-46)."""
    if not line.startswith("S"):
        raise ValueError("Given GFA2 line is not a segment.")
    parts = line.strip().split("\t")
    _ = parts[3]                       # the reference indexes the sequence field (IndexError if absent)
    return parts[1].strip(), int(parts[2])


def gfa2_parse_edge(line: str):
    """``E * <sid1> <sid2> <b1> <e1> <b2> <e2> <alignment>`` -> (sid1, sid2, arange, brange);
    mirrors gfa2_parse_edge (gfa.py:72-87)."""
    if not line.startswith("E"):
        raise ValueError("Given GFA2 line is not an edge.")
    parts = line.strip().split("\t")
    _ = parts[8]
    arange = tuple(map(_gfa_pos_to_int, parts[4:6]))
    brange = tuple(map(_gfa_pos_to_int, parts[6:8]))
    return parts[2].strip(), parts[3].strip(), arange, brange


def read_gfa2_rows(lines):
    """Pure-Python reading of a GFA2 text the way ``phasm layout`` does (assembler.py:56-60, 96-98):
    -> (names, lengths, rows) with node index 2*i for ``name+`` and 2*i+1 for ``name-``.  The bulk path
    is the native ``po_add_gfa``; this is the readable counterpart the tests compare it with."""
    lines = list(lines)
    order, length = {}, []
    for ln in lines:
        if ln.startswith("S"):
            sid, n = gfa2_parse_segment(ln)
            if sid in order:
                length[order[sid]] = n      # dict semantics: the last S line of a name wins (gfa.py:109)
            else:
                order[sid] = len(length)
                length.append(n)
    rows = []
    for ln in lines:
        if ln.startswith("E"):
            s1, s2, ar, br = gfa2_parse_edge(ln)
            nodes = []
            for sid in (s1, s2):
                if sid[-1:] not in ("+", "-"):
                    raise ValueError("segment reference without strand: %r" % sid)
                nodes.append(2 * order[sid[:-1]] + (sid[-1] == "-"))   # KeyError like reads[sid[:-1]] (gfa.py:97-98)
            rows.append((nodes[0], nodes[1], ar[0], ar[1], br[0], br[1]))
    names = list(order)
    return names, np.asarray(length, dtype=np.int64), np.asarray(rows, dtype=np.int64).reshape(-1, 6)
