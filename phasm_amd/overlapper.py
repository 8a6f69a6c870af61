"""Drop-in for the reference's ``phasm.overlapper`` extension module.

The reference exposes, through pybind11 (/root/reference/src/phasm.cpp:8-18)::

    class ExactOverlapper:
        def __init__(self)                       # src/overlapper.cpp:19
        def add_sequence(self, id: str, seq: str) -> None      # src/overlapper.cpp:22-26
        def overlaps(self, min_length: int) -> List[Tuple[str, str, int, int, int, int]]
                                                                # src/overlapper.cpp:28-150

and ``phasm overlap`` is its only caller (phasm/cli/assembler.py:31,39-42).  This class has
the same three methods with the same argument meaning, return type and error behaviour
(negative ``min_length`` -> ``TypeError`` like pybind11's unsigned conversion; ``bytes`` ids
and sequences accepted like pybind11's ``std::string`` caster), and calls the HIP library
through the C ABI of include/phasm_overlap.h.  Extra, non-reference methods
(``overlaps_array``, ``overlaps_shard_array``, ``stats``) expose the bulk 24-byte row array
so callers that want throughput do not have to build millions of Python tuples.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import CAND_DTYPE, EDGE_DTYPE, ROW_DTYPE, PoLayoutParams, PoLayoutStats, PoStats

OverlapT = Tuple[str, str, int, int, int, int]

_PYT = [False, None]


def _pytuples():
    """``po_rows_to_tuples`` of phasm_amd/_pytuples.so (built by phasm_amd.build.build_pytuples), or None."""
    if not _PYT[0]:
        _PYT[0] = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pytuples.so")
        if os.path.exists(path) and not os.environ.get("PHASM_NO_PYTUPLES"):
            try:
                lib = ctypes.PyDLL(path)
                lib.po_rows_to_tuples.restype = ctypes.py_object
                lib.po_rows_to_tuples.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.py_object]
                _PYT[1] = lib.po_rows_to_tuples
            except (OSError, AttributeError):
                _PYT[1] = None
    return _PYT[1]


def _to_bytes(x, what: str) -> bytes:
    if isinstance(x, bytes):
        return x
    if isinstance(x, str):
        return x.encode("utf-8")
    if isinstance(x, (bytearray, memoryview)):
        return bytes(x)
    raise TypeError("%s must be str or bytes, not %s" % (what, type(x).__name__))


def _check(handle, status: int):
    if status == _lib.PO_OK:
        return
    msg = _lib.load().po_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else "libphasm_overlap error %d" % status
    if status == _lib.PO_ERR_NOMEM:
        raise MemoryError(msg)
    if status == _lib.PO_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


class OverlapResult:
    """Owns one ``po_result``: rows stay on the device until asked for."""

    def __init__(self, owner: "ExactOverlapper", ptr, dtype=ROW_DTYPE):
        self._owner = owner
        self._ptr = ptr
        self._dtype = dtype  # ROW_DTYPE (24-byte rows) or CAND_DTYPE (16-byte verified candidates)
        self._lib = _lib.load()

    def __len__(self) -> int:
        return int(self._lib.po_result_count(self._ptr)) if self._ptr else 0

    def rows(self) -> np.ndarray:
        """Host copy as a structured array (a_idx, b_idx, astart, aend, bstart, bend)."""
        n = len(self)
        if n == 0:
            return np.empty(0, dtype=self._dtype)
        p = self._lib.po_result_rows(self._ptr)
        if not p:
            _check(self._owner._h, _lib.PO_ERR_HIP)
        buf = (ctypes.c_char * (n * self._dtype.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=self._dtype).copy()

    def rows_view(self) -> np.ndarray:
        """The same rows WITHOUT the copy: a read-only view of the library's page-locked host buffer, valid until
        ``free()`` (the C ABI's ``po_result_rows`` pointer as it is)."""
        n = len(self)
        if n == 0:
            return np.empty(0, dtype=self._dtype)
        p = self._lib.po_result_rows(self._ptr)
        if not p:
            _check(self._owner._h, _lib.PO_ERR_HIP)
        buf = (ctypes.c_char * (n * self._dtype.itemsize)).from_address(p)
        v = np.frombuffer(buf, dtype=self._dtype)
        v.flags.writeable = False
        return v

    def rows_range_view(self, first: int, count: int) -> np.ndarray:
        """``po_result_rows_range``: rows [first, first + count) in page-locked host memory (a view, valid until the
        next call on the overlapper) -- one rank's share of a merged multi-GPU result."""
        if count <= 0:
            return np.empty(0, dtype=self._dtype)
        p = self._lib.po_result_rows_range(self._ptr, int(first), int(count))
        if not p:
            _check(self._owner._h, _lib.PO_ERR_HIP)
        buf = (ctypes.c_char * (count * self._dtype.itemsize)).from_address(p)
        v = np.frombuffer(buf, dtype=self._dtype)
        v.flags.writeable = False
        return v

    def write_gfa_edges(self, fileobj) -> int:
        """Native bulk writer of the GFA2 ``E`` lines (``po_write_gfa_edges``) to a real file."""
        fileobj.flush()
        n = ctypes.c_uint64()
        _check(self._owner._h, self._lib.po_write_gfa_edges(self._ptr, fileobj.fileno(), ctypes.byref(n)))
        return int(n.value)

    def device_ptr(self) -> int:
        return int(self._lib.po_result_device_rows(self._ptr) or 0)

    def copy_to_device(self, dst_ptr: int, count: Optional[int] = None) -> None:
        """Device-to-device copy of the first ``count`` entries (default: all) to ``dst_ptr``."""
        if count is None or count >= len(self):
            _check(self._owner._h, self._lib.po_result_copy_to_device(self._ptr, ctypes.c_void_p(dst_ptr)))
        elif count > 0:
            _check(self._owner._h, self._lib.po_result_copy_prefix_to_device(self._ptr, ctypes.c_void_p(dst_ptr), int(count)))

    def free(self) -> None:
        if self._ptr and self._owner._h:
            self._lib.po_result_free(self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ExactOverlapper:
    def __init__(self, device: Optional[int] = None):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        st = self._lib.po_create(ctypes.byref(h))
        if st != _lib.PO_OK:
            raise MemoryError("po_create failed")
        self._h = h
        self._results: List[OverlapResult] = []
        if device is not None:
            _check(self._h, self._lib.po_set_device(self._h, int(device)))

    # ---- the reference API -------------------------------------------------------------
    def add_sequence(self, id, seq) -> None:
        bid = _to_bytes(id, "id")
        bseq = _to_bytes(seq, "seq")
        _check(self._h, self._lib.po_add_sequence(self._h, bid, len(bid), bseq, len(bseq)))

    def add_sequence_ptr(self, id, address: int, length: int) -> None:
        """``po_add_sequence`` as the C ABI has it: the sequence is ``length`` bytes at ``address`` (copied, like every
        sequence: the caller's memory is never written -- the GPU tests hand in a read-only mapping to hold the library
        to that, tests/checker.py GuardedReads)."""
        bid = _to_bytes(id, "id")
        _check(self._h, self._lib.po_add_sequence(self._h, bid, len(bid), ctypes.c_char_p(int(address)), int(length)))

    def add_fasta(self, path: str, both_strands: bool = True) -> int:
        """Native FASTA ingest (``po_add_fasta``): every record as ``name+`` / sequence and ``name-`` /
        reverse complement, as ``phasm overlap`` adds them (assembler.py:38-40).  Returns the record count."""
        n = ctypes.c_uint64()
        _check(self._h, self._lib.po_add_fasta(self._h, os.fsencode(path), 1 if both_strands else 0, ctypes.byref(n)))
        return int(n.value)

    def write_gfa_segments(self, fileobj) -> int:
        """``S <name> <length> *`` for every read pair, written natively to a real file (``po_write_gfa_segments``)."""
        fileobj.flush()
        n = ctypes.c_uint64()
        _check(self._h, self._lib.po_write_gfa_segments(self._h, fileobj.fileno(), ctypes.byref(n)))
        return int(n.value)

    def overlaps(self, min_length: int) -> List[OverlapT]:
        res = self.overlaps_to_host_result(min_length)
        try:
            ids = self.ids()
            fn = _pytuples()
            if fn is not None and len(res):
                # the list of tuples built natively from the page-locked row array (phasm_amd/csrc/pytuples.c), as
                # pybind11 builds the reference's from std::vector<OverlapT> (src/phasm.cpp:15)
                p = self._lib.po_result_rows(res._ptr)
                if not p:   # (a failed device->host copy or page-locked allocation: raise, never hand NULL to the builder)
                    _check(self._h, _lib.PO_ERR_HIP)
                return fn(p, len(res), ids)
            arr = res.rows_view()
            a_ids = [ids[i] for i in arr["a_idx"].tolist()]
            b_ids = [ids[i] for i in arr["b_idx"].tolist()]
            return list(zip(a_ids, b_ids, arr["astart"].tolist(), arr["aend"].tolist(),
                            arr["bstart"].tolist(), arr["bend"].tolist()))
        finally:
            res.free()

    # ---- bulk / multi-GPU extensions ------------------------------------------------------
    @staticmethod
    def _min_length(min_length) -> int:
        if isinstance(min_length, bool) or not isinstance(min_length, (int, np.integer)):
            raise TypeError("overlaps(): incompatible function arguments: min_length must be an unsigned int")
        if min_length < 0 or min_length > 0xFFFFFFFF:
            raise TypeError("overlaps(): incompatible function arguments: min_length must be an unsigned int")
        return int(min_length)

    def overlaps_result(self, min_length: int, shard: int = 0, nshards: int = 1) -> OverlapResult:
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_overlaps_shard(self._h, m, int(shard), int(nshards), ctypes.byref(r)))
        return OverlapResult(self, r)

    def overlaps_to_host_result(self, min_length: int) -> OverlapResult:
        """``po_overlaps_to_host``: the rows land in page-locked host memory while later chunks are still being
        computed; ``rows_view()`` / ``rows()`` of the result cost no further copy from the device."""
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_overlaps_to_host(self._h, m, ctypes.byref(r)))
        return OverlapResult(self, r)

    def overlaps_ex_result(self, min_length: int, max_diff: int = 0, band: int = 0) -> OverlapResult:
        """``po_overlaps_ex``: banded seed-extension DP with up to ``max_diff`` differences (an extension beyond the
        exact reference; ``max_diff = 0`` gives the rows of ``overlaps``)."""
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_overlaps_ex(self._h, m, int(max_diff), int(band), ctypes.byref(r)))
        return OverlapResult(self, r)

    def overlaps_ex_array(self, min_length: int, max_diff: int = 0, band: int = 0) -> np.ndarray:
        res = self.overlaps_ex_result(min_length, max_diff, band)
        try:
            return res.rows()
        finally:
            res.free()

    def candidates_result(self, min_length: int, shard: int = 0, nshards: int = 1) -> OverlapResult:
        """Verified candidates of one a-side shard (``po_candidates_shard``): the compact form that
        travels between GPUs; ``expand_result`` turns the merged array into rows."""
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_candidates_shard(self._h, m, int(shard), int(nshards), ctypes.byref(r)))
        return OverlapResult(self, r, CAND_DTYPE)

    def candidates_result_into(self, min_length: int, shard: int, nshards: int, dst_ptr: int, capacity: int):
        """``po_candidates_shard_into``: (result, written) -- the candidates go straight to ``dst_ptr`` when they fit."""
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        w = ctypes.c_int()
        _check(self._h, self._lib.po_candidates_shard_into(self._h, m, int(shard), int(nshards), ctypes.c_void_p(dst_ptr),
                                                           int(capacity), ctypes.byref(w), ctypes.byref(r)))
        return OverlapResult(self, r, CAND_DTYPE), bool(w.value)

    # ---- sliced wide index (multi-GPU; phasm_amd/dist.py IndexExchange) -----------------------------------
    def index_slice_build(self, min_length: int, slice_: int, n_slices: int) -> Tuple[bool, int, int]:
        """``po_index_slice_build``: build sub-table ``slice_`` of ``n_slices``; (is_wide, slice_bits, chain_entries)."""
        m = self._min_length(min_length)
        w, b, e = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        _check(self._h, self._lib.po_index_slice_build(self._h, m, int(slice_), int(n_slices), ctypes.byref(w), ctypes.byref(b), ctypes.byref(e)))
        return bool(w.value), int(b.value), int(e.value)

    def index_chunk_bytes(self, slice_bits: int, chain_capacity: int) -> int:
        return int(self._lib.po_index_chunk_bytes(int(slice_bits), int(chain_capacity), None))

    def index_slice_export(self, dst_ptr: int, chain_capacity: int) -> None:
        _check(self._h, self._lib.po_index_slice_export(self._h, ctypes.c_void_p(dst_ptr), int(chain_capacity)))

    def candidates_result_indexed(self, min_length: int, shard: int, nshards: int, index_ptr: int, n_slices: int, slice_bits: int,
                                  chain_capacity: int, dst_ptr: int = 0, capacity: int = 0):
        """``po_candidates_shard_indexed``: the shard call on a gathered sliced index; (result, written)."""
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        w = ctypes.c_int()
        _check(self._h, self._lib.po_candidates_shard_indexed(self._h, m, int(shard), int(nshards), ctypes.c_void_p(index_ptr),
                                                              int(n_slices), int(slice_bits), int(chain_capacity),
                                                              ctypes.c_void_p(dst_ptr) if dst_ptr else None, int(capacity),
                                                              ctypes.byref(w), ctypes.byref(r)))
        return OverlapResult(self, r, CAND_DTYPE), bool(w.value)

    def overlaps_shard_indexed_array(self, min_length: int, shard: int, nshards: int, index_ptr: int, n_slices: int,
                                     slice_bits: int, chain_capacity: int) -> np.ndarray:
        m = self._min_length(min_length)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_overlaps_shard_indexed(self._h, m, int(shard), int(nshards), ctypes.c_void_p(index_ptr),
                                                            int(n_slices), int(slice_bits), int(chain_capacity), ctypes.byref(r)))
        res = OverlapResult(self, r)
        try:
            return res.rows()
        finally:
            res.free()

    def expand_result(self, cand_device_ptr: int, n_candidates: int) -> OverlapResult:
        """Rows from a candidate array resident on this handle's device (``po_expand``)."""
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_expand(self._h, ctypes.c_void_p(cand_device_ptr), int(n_candidates), ctypes.byref(r)))
        return OverlapResult(self, r)

    def overlaps_array(self, min_length: int) -> np.ndarray:
        res = self.overlaps_to_host_result(min_length)
        try:
            return res.rows()
        finally:
            res.free()

    def overlaps_shard_array(self, min_length: int, shard: int, nshards: int) -> np.ndarray:
        res = self.overlaps_result(min_length, shard, nshards)
        try:
            return res.rows()
        finally:
            res.free()

    def shard_range(self, shard: int, nshards: int) -> Tuple[int, int]:
        """Read-index range scanned on the a-side by shard ``shard`` of ``nshards`` (host logic)."""
        b, e = ctypes.c_uint32(), ctypes.c_uint32()
        _check(self._h, self._lib.po_shard_range(self._h, int(shard), int(nshards), ctypes.byref(b), ctypes.byref(e)))
        return b.value, e.value

    def upload(self) -> None:
        _check(self._h, self._lib.po_upload(self._h))

    def upload_piece(self, shard: int, nshards: int, dst_ptr: int = 0, capacity_words: int = 0) -> Tuple[bool, int]:
        """``po_upload_piece``: (ok, words of the piece); with ``dst_ptr`` the piece is copied host->device there."""
        n, ok = ctypes.c_uint64(), ctypes.c_int()
        _check(self._h, self._lib.po_upload_piece(self._h, int(shard), int(nshards), ctypes.c_void_p(dst_ptr) if dst_ptr else None,
                                                  int(capacity_words), ctypes.byref(n), ctypes.byref(ok)))
        return bool(ok.value), int(n.value)

    def upload_assemble(self, pieces_ptr: int, slot_words: int, nshards: int, nparts: int = 1) -> None:
        _check(self._h, self._lib.po_upload_assemble_parts(self._h, ctypes.c_void_p(pieces_ptr), int(slot_words), int(nshards), int(nparts)))

    def upload_piece_part(self, shard: int, nshards: int, part: int, nparts: int, dst_ptr: int = 0, capacity_words: int = 0) -> Tuple[bool, int]:
        """``po_upload_piece_part``: part ``part`` of ``nparts`` of shard ``shard``'s piece (see ``upload_piece``)."""
        n, ok = ctypes.c_uint64(), ctypes.c_int()
        _check(self._h, self._lib.po_upload_piece_part(self._h, int(shard), int(nshards), int(part), int(nparts),
                                                       ctypes.c_void_p(dst_ptr) if dst_ptr else None, int(capacity_words),
                                                       ctypes.byref(n), ctypes.byref(ok)))
        return bool(ok.value), int(n.value)

    def invalidate(self) -> None:
        """``po_invalidate``: the next upload / overlaps call copies the packed reads to the device again."""
        _check(self._h, self._lib.po_invalidate(self._h))

    # ---- the consumer side: stage 1 of `phasm layout` (phasm_amd/layout.py is the host mirror) ----
    def add_segment(self, name, length: int) -> None:
        """One GFA2 ``S`` line without sequence (``po_add_segment``): nodes ``name+`` and ``name-``."""
        b = _to_bytes(name, "name")
        _check(self._h, self._lib.po_add_segment(self._h, b, len(b), int(length)))

    def add_gfa(self, path: str) -> Tuple[int, OverlapResult]:
        """Read a GFA2 file into this (empty) handle: segments become sequence-less reads, ``E`` lines
        the returned row result (``po_add_gfa``).  Returns (segment count, rows)."""
        n = ctypes.c_uint64()
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_add_gfa(self._h, os.fsencode(path), ctypes.byref(n), ctypes.byref(r)))
        return int(n.value), OverlapResult(self, r)

    def result_from_rows(self, rows: np.ndarray) -> OverlapResult:
        """Wrap caller-supplied rows (structured ROW_DTYPE or int array [n, 6]) into a result."""
        if rows.dtype != ROW_DTYPE:
            arr = np.asarray(rows)
            out = np.empty(len(arr), dtype=ROW_DTYPE)
            for k, name in enumerate(ROW_DTYPE.names):
                out[name] = arr[:, k]
            rows = out
        rows = np.ascontiguousarray(rows)
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_result_from_rows(self._h, rows.ctypes.data_as(ctypes.c_void_p), len(rows), ctypes.byref(r)))
        return OverlapResult(self, r)

    def layout_edges(self, rows: OverlapResult, min_read_length: int = 0, min_overlap_length: int = 0,
                     max_overhang_abs: int = 1000, max_overhang_rel: float = 0.8,
                     want_removed: bool = True) -> Tuple[OverlapResult, Optional[np.ndarray]]:
        """``po_layout_edges``: assembly-graph edges (EDGE_DTYPE result) and the contained-read flags."""
        prm = PoLayoutParams(int(min_read_length), int(min_overlap_length), int(max_overhang_abs), 0, float(max_overhang_rel))
        removed = np.zeros(len(self) // 2, dtype=np.uint8) if want_removed else None
        r = ctypes.c_void_p()
        _check(self._h, self._lib.po_layout_edges(
            self._h, rows._ptr, ctypes.byref(prm),
            removed.ctypes.data_as(ctypes.c_void_p) if removed is not None and len(removed) else None, ctypes.byref(r)))
        return OverlapResult(self, r, EDGE_DTYPE), removed

    def layout_stats(self) -> dict:
        s = PoLayoutStats()
        _check(self._h, self._lib.po_get_layout_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def __len__(self) -> int:
        return int(self._lib.po_num_sequences(self._h))

    def ids(self) -> List[str]:
        out = []
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        for i in range(len(self)):
            self._lib.po_get_id(self._h, i, ctypes.byref(p), ctypes.byref(n))
            out.append(ctypes.string_at(p.value, n.value).decode("utf-8", "surrogateescape") if n.value else "")
        return out

    def lengths(self) -> np.ndarray:
        return np.array([self._lib.po_get_length(self._h, i) for i in range(len(self))], dtype=np.int64)

    def stats(self) -> dict:
        s = PoStats()
        _check(self._h, self._lib.po_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.po_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
