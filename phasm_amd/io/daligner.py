"""DAZZ_DB ``DBdump`` / DALIGNER ``LAdump`` text -> the overlap file's wire format (GFA2 ``S`` / ``E`` lines).

The other producer of the file that ``phasm layout`` reads.  Counterpart of

* ``parse_reads``              /root/reference/phasm/io/daligner.py:51-108
* ``parse_local_alignments``   /root/reference/phasm/io/daligner.py:111-164
* ``full_id``                  /root/reference/phasm/io/daligner.py:46-48
* ``generate_moviename_hash``  /root/reference/phasm/io/daligner.py:19-29
* ``daligner2gfa``             /root/reference/phasm/cli/convert.py:65-133

Same record keys, the same line dispatch (a line is recognised by its first character; a trace-point line by
two leading blanks; the ``+``/``@``/``%`` size lines of a dump are skipped because they match nothing) and the
same exceptions on malformed lines.  ``write_gfa`` gives byte-identical ``H``/``S``/``E`` lines; the one
deliberate difference is ``with_sequences``: the reference passes a 0-d numpy bytes array through ``str()``
there (convert.py:97-99 with daligner.py:103), which prints a numpy-version-dependent repr instead of bases --
here the bases are written.  ``to_rows`` is the bulk form for the device path: read names, lengths and the
24-byte row array (node 2i = ``name+``, 2i+1 = ``name-``), with no text in between.
"""
from __future__ import annotations

import enum
import hashlib
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from . import gfa


class Strand(enum.IntEnum):
    SAME = 0
    OPPOSITE = 1


def generate_moviename_hash(filename: str) -> str:
    """First eight bytes of SHA-256(filename), little endian, in decimal (daligner.py:19-29)."""
    return str(int.from_bytes(hashlib.sha256(filename.encode("utf-8")).digest()[:8], byteorder="little"))


def pacbio_name(moviename: str, index: int, length: int) -> str:
    """The header ``fix_header`` gives the ``index``-th (0-based) read (daligner.py:37-40)."""
    return "m000000_000000_00000_c%s/%d/0_%d" % (moviename, index + 1, length)


def full_id(read: dict) -> str:
    return "%s/%s/%s_%s" % (read["moviename"], read["read_id"], read["pulse_start"], read["pulse_end"])


def _want(parts, n, what, line):
    if len(parts) != n:
        raise ValueError("Unexpected input when reading %s from DAZZ_DB: expected %d fields. Line: %r" % (what, n, line))


def parse_reads(input_stream: Iterable[str]) -> Iterator[dict]:
    """One dict per read of a ``DBdump -rh`` text: read_id, moviename, pulse_start, pulse_end, length, well,
    and ``sequence`` (here a ``str``) when the dump has ``S`` lines."""
    cur: dict = {}
    for line in input_stream:
        c = line[:1]
        if c not in "RHLS" or not c:
            continue
        parts = line.split()
        if c == "R":
            _want(parts, 2, "read ID", line)
            if cur:
                yield cur
                cur = {}
            cur["read_id"] = parts[1]
        elif c == "H":
            _want(parts, 3, "reads", line)
            cur["moviename"] = parts[2]
        elif c == "L":
            _want(parts, 4, "read lengths", line)
            ps, pe = int(parts[2]), int(parts[3])
            well = int(parts[1])
            cur.update(pulse_start=ps, pulse_end=pe, length=pe - ps, well=well)
        else:
            cur["sequence"] = parts[2].strip()      # IndexError when the field is missing, as in the reference
    if cur:
        yield cur


def parse_local_alignments(input_stream: Iterable[str]) -> Iterator[dict]:
    """One dict per ``P`` record of an ``LAdump -cdt`` text: a, b, strand, arange, brange and, when present,
    trace_points (list of (differences, b-length) pairs) and differences."""
    cur: dict = {}
    expect = seen = 0
    for line in input_stream:
        c = line[:1]
        if c == "P":
            parts = line.split()
            if cur and "a" in cur and "b" in cur:
                yield cur
                cur = {}
                expect = seen = 0
            cur["a"], cur["b"] = parts[1], parts[2]
            cur["strand"] = Strand.SAME if parts[3] == "n" else Strand.OPPOSITE
        elif c == "C":
            a_start, a_end, b_start, b_end = map(int, line.split()[1:])
            cur["arange"] = (a_start, a_end)
            cur["brange"] = (b_start, b_end)
        elif c == "T":
            cur["trace_points"] = []
            expect = int(line.split()[1])
            seen = 0
        elif line.startswith("  "):
            if seen >= expect:
                raise ValueError("Received more tracepoints than expected (expected %d tracepoints)." % expect)
            cur["trace_points"].append(tuple(map(int, line.split())))
            # the reference never advances its counter either (daligner.py:121-157): the check above only
            # fires for a record announced with `T 0`
        elif c == "D":
            cur["differences"] = int(line.split()[1])
    if cur and "a" in cur and "b" in cur:
        yield cur


def _external_id(read: dict, translations: Optional[Dict[str, str]]) -> str:
    rid = read["read_id"]
    if translations:
        rid = translations[full_id(read)]
    return rid.split()[0]            # only the part before the first blank (convert.py:90-91)


def write_gfa(out, db_input: Iterable[str], las_input: Iterable[str], with_sequences: bool = False,
              with_trace_points: Optional[int] = None, translations: Optional[Dict[str, str]] = None) -> Tuple[int, int]:
    """``daligner2gfa`` (convert.py:65-133): header (with ``TS:i:<spacing>`` when trace points are asked for), one
    ``S`` line per read, one ``E`` line per local alignment -- a on ``+``, b on ``+`` (same strand) or ``-``, an end
    position that equals the read length marked with ``$``.  Returns (#S lines, #E lines)."""
    out.write(gfa.gfa_header(trace_spacing=with_trace_points))
    ext: Dict[str, str] = {}
    length: Dict[str, int] = {}
    n_s = n_e = 0
    for read in parse_reads(db_input):
        rid = _external_id(read, translations)
        ext[read["read_id"]] = rid
        length[read["read_id"]] = read["length"]
        seq = read["sequence"] if with_sequences and "sequence" in read else "*"
        out.write("S\t%s\t%d\t%s\n" % (rid, read["length"], seq))
        n_s += 1
    for la in parse_local_alignments(las_input):
        a, b = ext[la["a"]], ext[la["b"]]
        (s, e), (bs, be) = la["arange"], la["brange"]
        if with_trace_points and "trace_points" in la:
            tp = ",".join(str(t[1]) for t in la["trace_points"])
        else:
            tp = "*"
        out.write("E\t*\t%s+\t%s%s\t%d\t%d%s\t%d\t%d%s\t%s\n" % (
            a, b, "+" if la["strand"] == Strand.SAME else "-",
            s, e, "$" if e == length[la["a"]] else "", bs, be, "$" if be == length[la["b"]] else "", tp))
        n_e += 1
    return n_s, n_e


def to_rows(db_input: Iterable[str], las_input: Iterable[str],
            translations: Optional[Dict[str, str]] = None) -> Tuple[List[str], np.ndarray, np.ndarray]:
    """What ``phasm layout`` would see after ``daligner2gfa`` and its own GFA2 reader, without the text:
    (names, lengths, rows[n, 6] = a_node, b_node, astart, aend, bstart, bend).  Two dump reads that end up with the
    same external name share a node pair and the later length wins, as with repeated ``S`` lines (gfa.py:109)."""
    order: Dict[str, int] = {}
    lengths: List[int] = []
    node: Dict[str, int] = {}
    for read in parse_reads(db_input):
        rid = _external_id(read, translations)
        k = order.get(rid)
        if k is None:
            k = order[rid] = len(lengths)
            lengths.append(read["length"])
        else:
            lengths[k] = read["length"]
        node[read["read_id"]] = 2 * k
    flat: List[int] = []
    for la in parse_local_alignments(las_input):
        a, b = node[la["a"]], node[la["b"]] + int(la["strand"])
        flat.extend((a, b) + la["arange"] + la["brange"])
    return list(order), np.asarray(lengths, dtype=np.int64), np.asarray(flat, dtype=np.int64).reshape(-1, 6)
