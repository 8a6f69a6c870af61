"""The host half of po_overlaps_to_host's compact row transfer (c_api.hip, namespace home), on the CPU.

The device hands out one 16-byte record {a, p, b, type} per verified candidate -- in paired-strand mode one per strand-mirror
pair -- and the library's helper threads write the 24-byte rows.  The rows must be, in this order, what the device-side
emission (kernels.hip.h write_rows) writes; the rules are restated here in plain Python from the row definition
(/root/reference/src/overlapper.cpp:77-82 A rows, :104-110 B rows) and the mirror rules of SURVEY.md section 8c, and held
against the reference's own goldens: every golden row must come out of the records derived from the goldens."""
import ctypes

import numpy as np
import pytest

import golden_utils as gu
from phasm_amd import _lib
from phasm_amd._lib import CAND_DTYPE, ROW_DTYPE


def expand_py(recs, lens, paired):
    """write_rows, one record at a time: A row, [its mirror], B row, [its mirror]."""
    out = []
    for a, p, b, t in recs:
        la, lb = int(lens[a]), int(lens[b])
        if t & 1:
            l = la - p
            out.append((a, b, p, la, 0, l))
            if paired and a != (b ^ 1):
                out.append((b ^ 1, a ^ 1, lb - l, lb, 0, l))
        if t & 2:
            out.append((a, b, p, p + lb, 0, lb))
            if paired:
                out.append((a ^ 1, b ^ 1, la - p - lb, la - p, 0, lb))
    return out


def pack_shifts(n_reads, longest):
    """The widths po_overlaps_to_host picks (c_api.hip, home_begin): a and b in ceil(log2 reads) bits, p above them."""
    br = 1
    while (1 << br) < n_reads:
        br += 1
    bp = 1
    while (1 << bp) < longest + 1:
        bp += 1
    assert 2 * br + bp <= 62
    return br, 2 * br


PACKED = False   # (set by the fixture below: every test of this file runs with both record forms)


@pytest.fixture(autouse=True, params=["16-byte records", "8-byte records"])
def record_form(request):
    global PACKED
    PACKED = request.param.startswith("8")
    yield
    PACKED = False


def expand_lib(recs, lens, paired, n_rows=None):
    lib = _lib.load()
    L = np.ascontiguousarray(lens, dtype=np.uint32)
    want = len(expand_py(recs, lens, paired)) if n_rows is None else n_rows
    out = np.empty(want + 2, dtype=ROW_DTYPE)
    out.view(np.uint8)[:] = 0x5A
    if PACKED:
        # (a foreign read index must still fit its field to travel at all: widths from the largest index named)
        top = max([len(L)] + [max(a, b) + 1 for a, _, b, _ in recs])
        sh_b, sh_p = pack_shifts(top, int(L.max()) if len(L) else 1)
        r8 = np.array([a | (b << sh_b) | (p << sh_p) | (t << 62) for a, p, b, t in recs], dtype=np.uint64)
        rc = lib.po_debug_expand_packed(r8.ctypes.data_as(ctypes.c_void_p), len(r8), sh_b, sh_p, L.ctypes.data_as(ctypes.c_void_p), len(L),
                                        1 if paired else 0, out.ctypes.data_as(ctypes.c_void_p), want)
        return rc, out
    r = np.zeros(len(recs), dtype=CAND_DTYPE)
    if len(recs):
        arr = np.asarray(recs, dtype=np.uint32).reshape(-1, 4)
        for k, name in enumerate(CAND_DTYPE.names):
            r[name] = arr[:, k]
    rc = lib.po_debug_expand_records(r.ctypes.data_as(ctypes.c_void_p), len(r), L.ctypes.data_as(ctypes.c_void_p), len(L),
                                     1 if paired else 0, out.ctypes.data_as(ctypes.c_void_p), want)
    return rc, out


def test_packed_records_keep_full_width_fields():
    """The widest read set one word can hold: 2^24 reads of up to 2^14 - 1 bases, and 2^17 reads of up to 2^28 - 1."""
    lib = _lib.load()
    for n_reads, longest in ((1 << 24, (1 << 14) - 1), (1 << 17, (1 << 28) - 1)):
        sh_b, sh_p = pack_shifts(n_reads, longest)
        lens = np.full(n_reads, longest, dtype=np.uint32)
        a, b, p = n_reads - 2, n_reads - 1, longest - 7
        r8 = np.array([a | (b << sh_b) | (p << sh_p) | (1 << 62)], dtype=np.uint64)
        out = np.empty(2, dtype=ROW_DTYPE)
        rc = lib.po_debug_expand_packed(r8.ctypes.data_as(ctypes.c_void_p), 1, sh_b, sh_p, lens.ctypes.data_as(ctypes.c_void_p), n_reads, 0,
                                        out.ctypes.data_as(ctypes.c_void_p), 1)
        assert rc == 0
        assert rows_list(out[:1]) == [(a, b, p, longest, 0, 7)]
    assert lib.po_debug_expand_packed(r8.ctypes.data_as(ctypes.c_void_p), 1, 0, 0, lens.ctypes.data_as(ctypes.c_void_p), n_reads, 0,
                                      out.ctypes.data_as(ctypes.c_void_p), 1) == -2


def rows_list(arr):
    return [tuple(int(x[n]) for n in ROW_DTYPE.names) for x in arr]


@pytest.mark.parametrize("paired", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 7, 8191, 8192, 8193, 50_000])
def test_rows_of_random_records(n, paired):
    rng = np.random.default_rng(100 + n + paired)
    n_reads = 40
    lens = rng.integers(50, 4000, size=n_reads).astype(np.uint32)
    if paired:
        lens[1::2] = lens[0::2]
    recs = []
    for _ in range(n):
        a = int(rng.integers(0, n_reads))
        b = int(rng.integers(0, n_reads - 1))
        b += b >= a
        if paired and rng.random() < 0.05:
            b = a ^ 1                      # a read against its own reverse complement: the A row is its own mirror
        t = int(rng.integers(1, 4))
        la, lb = int(lens[a]), int(lens[b])
        if t & 2:                           # b inside a
            if lb > la:
                a, b, la, lb = b, a, lb, la
            p = la - lb if (t & 1) else int(rng.integers(0, la - lb + 1))
        else:
            p = int(rng.integers(max(0, la - lb), la))
        recs.append((a, p, b, t))
    want = expand_py(recs, lens, paired)
    rc, out = expand_lib(recs, lens, paired)
    assert rc == 0
    assert rows_list(out[:len(want)]) == want
    assert bytes(out[len(want):].view(np.uint8)) == b"\x5a" * (2 * ROW_DTYPE.itemsize)   # nothing written behind the rows


def test_a_wrong_row_count_and_a_foreign_read_are_errors():
    lens = np.array([100, 100, 80, 80], dtype=np.uint32)
    recs = [(0, 40, 2, 1), (2, 10, 1, 1)]
    rc, _ = expand_lib(recs, lens, 1, n_rows=5)      # the records give 4 rows
    assert rc == 1
    rc, _ = expand_lib([(0, 40, 9, 1)], lens, 0, n_rows=1)
    assert rc == 2
    rc, out = expand_lib(recs, lens, 1)               # (the pool works again after an error)
    assert rc == 0 and rows_list(out[:4]) == expand_py(recs, lens, 1)


def test_records_derived_from_the_reference_goldens_give_back_the_goldens():
    """Ladder goldens are outputs of the reference on (x, revcomp x) read sets: keep the canonical member of every
    strand-mirror pair as a record, expand, and the reference's multiset must come back."""
    for name in ("ladder_small", "ladder_varlen", "cfg2_1k"):
        _, seqs, m, want = gu.ladder_case(name)
        lens = np.array([len(s) for s in seqs], dtype=np.uint32)
        from collections import Counter
        cnt = Counter(map(tuple, want.tolist()))
        recs = {}
        for (a, b, s, e, bs, be), k in cnt.items():
            la, lb = int(lens[a]), int(lens[b])
            is_a = e == la                    # suffix of a = prefix of b
            is_b = be == lb                   # all of b inside a
            assert is_a or is_b
            if is_a and is_b:
                assert k == 2                 # the A + B duplicate (overlapper.cpp:77-82 and :104-110 both fire)
            # canonical member: A rows with a <= b^1 (index order), B rows with a on the + strand
            if is_a:
                if a <= (b ^ 1):
                    recs[(a, s, b)] = recs.get((a, s, b), 0) | 1
            if is_b:
                if a % 2 == 0:
                    recs[(a, s, b)] = recs.get((a, s, b), 0) | 2
        rl = [(a, p, b, t) for (a, p, b), t in sorted(recs.items())]
        rc, out = expand_lib(rl, lens, 1)
        assert rc == 0
        got = Counter(rows_list(out[:len(out) - 2]))
        assert got == cnt, name
