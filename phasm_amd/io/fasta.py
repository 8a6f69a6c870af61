"""FASTA ingest + reverse complement for the ``overlap`` command.

The reference delegates both to the third-party ``dinopy`` package, which is not vendored
(``dinopy.FastaReader(path).entries()`` and ``dinopy.reverse_complement``:
/root/reference/phasm/cli/assembler.py:32,35-37,40; pinned only as ``dinopy>=2.0`` in
requirements.txt:4).  This is a small self-contained replacement with the behaviour the
call site relies on:

* the entry name is the *whole* header line after ``>`` (assembler.py:36 uses
  ``entry.name`` undivided), sequence lines are concatenated, blank lines are skipped
  (tests/data/test.fasta in the reference has blank lines between records);
* the sequence is passed through as-is -- no upper-casing (assembler.py:37);
* ``reverse_complement`` complements IUPAC letters and preserves case; for pure ``ACGT``
  this is unambiguous; other letters are "parity unpinned" (SURVEY.md section 8c).
"""
from __future__ import annotations

from typing import BinaryIO, Iterator, Tuple, Union

_COMP = bytes.maketrans(b"ACGTURYKMBVDHSWNacgturykmbvdhswn",
                        b"TGCAAYRMKVBHDSWNtgcaayrmkvbhdswn")


def reverse_complement(seq: Union[bytes, str]) -> Union[bytes, str]:
    if isinstance(seq, str):
        return seq.encode("latin-1").translate(_COMP)[::-1].decode("latin-1")
    return bytes(seq).translate(_COMP)[::-1]


def read_fasta(f: Union[str, BinaryIO]) -> Iterator[Tuple[str, bytes]]:
    """Yield ``(name, sequence_bytes)`` for every record."""
    close = False
    if isinstance(f, (str, bytes)):
        f = open(f, "rb")
        close = True
    try:
        name = None
        chunks = []
        for line in f:
            if isinstance(line, str):
                line = line.encode("latin-1")
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(chunks)
                name = line[1:].decode("utf-8")
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
        if name is not None:
            yield name, b"".join(chunks)
    finally:
        if close:
            f.close()
