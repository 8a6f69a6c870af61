"""The C-ABI library loads and exports every symbol include/phasm_overlap.h declares; host-side
logic that needs no GPU behaves; compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from phasm_amd import _lib
from phasm_amd.overlapper import ExactOverlapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "phasm_overlap.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(po_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(s[0] for s in _lib.SYMBOLS) == names  # the ctypes table binds exactly the header
    assert lib.po_abi_version() == 4


def test_row_struct_is_24_bytes():
    assert _lib.ROW_DTYPE.itemsize == 24
    assert ctypes.sizeof(_lib.PoStats) == 4 * 4 + 10 * 8 + 10 * 4 + 16 + 24 + 8 + 8 + 8 + 8


def test_host_side_store_and_shard_ranges():
    ov = ExactOverlapper()
    seqs = ["ACGT" * k for k in range(1, 11)]
    for i, s in enumerate(seqs):
        ov.add_sequence("read%d+" % i, s)
    ov.add_sequence(b"bytes-id", b"ACGTNNNN")      # bytes accepted like pybind11's std::string caster
    assert len(ov) == 11
    assert ov.ids()[3] == "read3+" and ov.ids()[-1] == "bytes-id"
    assert ov.lengths().tolist() == [4 * k for k in range(1, 11)] + [8]
    # shards: contiguous, cover everything, balanced by bases
    for ns in (1, 2, 3, 8, 20):
        rng = [ov.shard_range(k, ns) for k in range(ns)]
        assert rng[0][0] == 0 and rng[-1][1] == 11
        assert all(rng[k][1] == rng[k + 1][0] for k in range(ns - 1))
    with pytest.raises(ValueError):
        ov.shard_range(3, 3)
    with pytest.raises(TypeError):
        ov.overlaps(-1)
    with pytest.raises(TypeError):
        ov.overlaps("10")
    with pytest.raises(TypeError):
        ov.add_sequence(5, "ACGT")
    ov.close()


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_overlaps_fails_loudly_without_gpu():
    ov = ExactOverlapper()
    ov.add_sequence("a", "ACGTACGT")
    ov.add_sequence("b", "ACGTACGT")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback|hip"):
        ov.overlaps(3)
    ov.close()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under phasm_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "phasm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "overlap_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_importable_under_the_reference_module_name():
    """`from phasm.overlapper import ExactOverlapper` (assembler.py:15) works with the overlay."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    try:
        mod = importlib.import_module("phasm.overlapper")
        assert mod.ExactOverlapper is ExactOverlapper
        ov = mod.ExactOverlapper()
        assert {"add_sequence", "overlaps"} <= set(dir(ov))
        ov.close()
    finally:
        sys.path.remove(os.path.join(ROOT, "integration"))
        sys.modules.pop("phasm.overlapper", None)
        sys.modules.pop("phasm", None)


def test_sliced_index_arguments_are_checked_at_the_boundary():
    """The *_indexed entry points take a (n_slices, slice_bits, chain_capacity) triple that describes a buffer the
    library cannot see: what cannot be a sliced index is refused before anything is computed or addressed (a shift by
    slice_bits >= 64 is undefined; the scan addresses 2^slice_bits slots per chunk)."""
    lib = _lib.load()
    assert lib.po_index_chunk_bytes(0, 10, None) == 0 and lib.po_index_chunk_bytes(31, 10, None) == 0
    assert lib.po_index_chunk_bytes(64, 10, None) == 0 and lib.po_index_chunk_bytes(70, 10, None) == 0
    off = ctypes.c_uint64()
    assert lib.po_index_chunk_bytes(10, 100, ctypes.byref(off)) == ((1025 * 16 + 255) // 256 * 256) + (800 + 255) // 256 * 256
    assert off.value == (1025 * 16 + 255) // 256 * 256
    h = ctypes.c_void_p()
    assert lib.po_create(ctypes.byref(h)) == 0
    buf = (ctypes.c_char * 4096)()
    out = ctypes.c_void_p()
    written = ctypes.c_int()
    for n_slices, bits, cap in ((2, 0, 8), (2, 31, 8), (2, 64, 8), (2, 200, 8), (1, 10, 8), (5000, 10, 8), (2, 10, 1 << 40)):
        st = lib.po_candidates_shard_indexed(h, 100, 0, 2, ctypes.cast(buf, ctypes.c_void_p), n_slices, bits, cap, None, 0,
                                             ctypes.byref(written), ctypes.byref(out))
        assert st == _lib.PO_ERR_INVALID, (n_slices, bits, cap, st)
        st = lib.po_overlaps_shard_indexed(h, 100, 0, 2, ctypes.cast(buf, ctypes.c_void_p), n_slices, bits, cap, ctypes.byref(out))
        assert st == _lib.PO_ERR_INVALID, (n_slices, bits, cap, st)
    lib.po_destroy(h)


def test_native_tuple_builder_matches_the_python_one():
    """ExactOverlapper.overlaps() returns the reference's list of 6-tuples (src/phasm.cpp:15); the list is built natively
    from the row array (phasm_amd/csrc/pytuples.c) -- same objects as the pure-Python form, shared id strings."""
    from phasm_amd import build, overlapper
    if build.build_pytuples() is None:
        pytest.skip("Python.h not available: the shim builds the tuples in Python")
    fn = overlapper._pytuples()
    assert fn is not None
    rng = np.random.default_rng(3)
    n = 5000
    arr = np.zeros(n, dtype=_lib.ROW_DTYPE)
    ids = ["read%d%s" % (i // 2, "+-"[i & 1]) for i in range(40)]
    arr["a_idx"], arr["b_idx"] = rng.integers(0, 40, n), rng.integers(0, 40, n)
    for f in ("astart", "aend", "bstart", "bend"):
        arr[f] = rng.integers(-5, 2_000_000_000, n)
    arr["bstart"][::2] = 0
    arr["aend"][::3] = rng.integers(0, 70000, len(arr["aend"][::3]))     # both sides of the shared-int table's 2^16 bound
    arr["astart"][:4] = (0, 65535, 65536, -1)
    got = fn(arr.ctypes.data, n, ids)
    want = list(zip([ids[i] for i in arr["a_idx"].tolist()], [ids[i] for i in arr["b_idx"].tolist()], arr["astart"].tolist(),
                    arr["aend"].tolist(), arr["bstart"].tolist(), arr["bend"].tolist()))
    assert got == want and all(type(t) is tuple and type(t[2]) is int for t in got[:50])
    assert got[0][0] is ids[int(arr["a_idx"][0])]          # the id strings are shared, not copied
    assert fn(arr.ctypes.data, 0, ids) == []
    arr["a_idx"][7] = 40
    with pytest.raises(IndexError):
        fn(arr.ctypes.data, n, ids)
