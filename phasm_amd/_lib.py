"""ctypes binding of libphasm_overlap.so (include/phasm_overlap.h).

There is no fallback: if the shared library is missing this raises ImportError, and the
library itself fails with PO_ERR_HIP when no GPU is usable.  (``cffi`` is not installed in the
target image, hence ctypes.)
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHASM_LIB") or os.path.join(HERE, "libphasm_overlap.so")  # PHASM_LIB: dev override

PO_OK, PO_ERR_INVALID, PO_ERR_NOMEM, PO_ERR_HIP, PO_ERR_CAPACITY = range(5)

ROW_DTYPE = np.dtype([("a_idx", "<u4"), ("b_idx", "<u4"), ("astart", "<i4"),
                      ("aend", "<i4"), ("bstart", "<i4"), ("bend", "<i4")])


CAND_DTYPE = np.dtype([("a_idx", "<u4"), ("p", "<u4"), ("b_idx", "<u4"), ("type", "<u4")])


EDGE_DTYPE = np.dtype([("u", "<u4"), ("v", "<u4"), ("weight", "<i4"), ("overlap_len", "<i4")])


class PoLayoutParams(ctypes.Structure):
    _fields_ = [("min_read_length", ctypes.c_uint32), ("min_overlap_length", ctypes.c_uint32),
                ("max_overhang_abs", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("max_overhang_rel", ctypes.c_double)]


class PoLayoutStats(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_uint64), ("n_type", ctypes.c_uint64 * 4), ("n_short", ctypes.c_uint64),
                ("n_min_overlap", ctypes.c_uint64), ("n_overhang", ctypes.c_uint64), ("n_pass", ctypes.c_uint64),
                ("n_contained_reads", ctypes.c_uint64), ("n_edges", ctypes.c_uint64),
                ("ms_classify", ctypes.c_float), ("ms_dedupe", ctypes.c_float), ("ms_emit", ctypes.c_float),
                ("ms_total", ctypes.c_float)]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["n_type"] = list(self.n_type)
        return d


class PoStats(ctypes.Structure):
    _fields_ = [
        ("bits_per_base", ctypes.c_uint32), ("kmer", ctypes.c_uint32),
        ("paired", ctypes.c_uint32), ("wide_index", ctypes.c_uint32),
        ("n_reads", ctypes.c_uint64), ("n_eligible", ctypes.c_uint64),
        ("total_bases", ctypes.c_uint64), ("shard_bases", ctypes.c_uint64),
        ("n_tiles", ctypes.c_uint64), ("n_candidates", ctypes.c_uint64),
        ("n_verified", ctypes.c_uint64), ("n_rows", ctypes.c_uint64),
        ("sum_overlap_bases", ctypes.c_uint64), ("verify_bytes_algo", ctypes.c_uint64),
        ("ms_index", ctypes.c_float), ("ms_scan_count", ctypes.c_float),
        ("ms_scan_fill", ctypes.c_float), ("ms_verify", ctypes.c_float),
        ("ms_select", ctypes.c_float), ("ms_emit", ctypes.c_float),
        ("ms_total", ctypes.c_float), ("ms_upload", ctypes.c_float),
        ("ms_scan_probe", ctypes.c_float), ("ms_verify_kernel", ctypes.c_float),
        ("verify_bytes_exec", ctypes.c_uint64), ("dp_steps", ctypes.c_uint64), ("dp_stopped", ctypes.c_uint64),
        ("max_diff", ctypes.c_uint32), ("band", ctypes.c_uint32), ("index_reused", ctypes.c_uint32),
        ("dp_lanes", ctypes.c_uint32), ("upload_bytes", ctypes.c_uint64),
        ("streamed", ctypes.c_uint32), ("n_deferred", ctypes.c_uint32),
        ("fused_tail", ctypes.c_uint32), ("tail_fallback", ctypes.c_uint32),
        ("n_predicted", ctypes.c_uint32), ("home_record_bytes", ctypes.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/phasm_overlap.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
SYMBOLS = [
    ("po_abi_version", ctypes.c_int, []),
    ("po_create", ctypes.c_int, [ctypes.POINTER(_P)]),
    ("po_destroy", None, [_P]),
    ("po_set_device", ctypes.c_int, [_P, ctypes.c_int]),
    ("po_add_sequence", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    ("po_add_fasta", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    ("po_num_sequences", ctypes.c_uint32, [_P]),
    ("po_get_id", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]),
    ("po_get_length", ctypes.c_uint32, [_P, ctypes.c_uint32]),
    ("po_upload", ctypes.c_int, [_P]),
    ("po_invalidate", ctypes.c_int, [_P]),
    ("po_upload_piece", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]),
    ("po_upload_assemble", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]),
    ("po_upload_piece_part", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                            ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]),
    ("po_upload_assemble_parts", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]),
    ("po_overlaps", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("po_overlaps_to_host", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("po_overlaps_ex", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("po_overlaps_shard", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("po_candidates_shard", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("po_expand", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(_P)]),
    ("po_candidates_shard_into", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                               ctypes.c_uint64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_P)]),
    ("po_index_slice_build", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]),
    ("po_index_chunk_bytes", ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    ("po_index_slice_export", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64]),
    ("po_candidates_shard_indexed", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                                   ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                   ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_P)]),
    ("po_overlaps_shard_indexed", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                                 ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(_P)]),
    ("po_shard_range", ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    ("po_result_count", ctypes.c_uint64, [_P]),
    ("po_result_rows", ctypes.c_void_p, [_P]),
    ("po_result_rows_range", ctypes.c_void_p, [_P, ctypes.c_uint64, ctypes.c_uint64]),
    ("po_result_device_rows", ctypes.c_void_p, [_P]),
    ("po_result_copy_to_device", ctypes.c_int, [_P, ctypes.c_void_p]),
    ("po_result_copy_prefix_to_device", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64]),
    ("po_result_free", None, [_P]),
    ("po_write_gfa_edges", ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    ("po_write_gfa_segments", ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    ("po_add_segment", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]),
    ("po_result_from_rows", ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(_P)]),
    ("po_add_gfa", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(_P)]),
    ("po_layout_edges", ctypes.c_int, [_P, _P, ctypes.POINTER(PoLayoutParams), ctypes.c_void_p, ctypes.POINTER(_P)]),
    ("po_get_layout_stats", ctypes.c_int, [_P, ctypes.POINTER(PoLayoutStats)]),
    ("po_get_stats", ctypes.c_int, [_P, ctypes.POINTER(PoStats)]),
    ("po_last_error", ctypes.c_char_p, [_P]),
    ("po_debug_fault_backtrace", ctypes.c_int, [ctypes.c_int]),
    ("po_debug_store_words", ctypes.c_uint64, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("po_debug_expand_packed", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64]),
    ("po_debug_expand_records", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_void_p, ctypes.c_uint64]),
    ("po_debug_host_ranges", ctypes.c_uint64, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64]),
    ("po_debug_pointer_info", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
]

_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own ``libamdhip64.so`` (SONAME
    ``libamdhip64.so.7``) and link to it by its unversioned file name, so a process that loads this
    library first (bound to /opt/rocm's copy) and torch afterwards ends up with two runtimes, and the second
    one finds no device.  Loading torch's copy first -- by path, without importing torch -- makes both bind
    to the same one, whatever the import order."""
    import importlib.util
    if os.environ.get("PHASM_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -m phasm_amd.build` (hipcc, gfx950). "
                "phasm_amd has no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib
