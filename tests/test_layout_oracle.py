"""The layout oracle (oracle/layout_oracle.py) against the golden vectors produced by the reference's own
classes (tests/golden/layout_cases.json), the order-independent form against the literal one, and the
native GFA2 reader (host logic of the C ABI, no GPU) against the Python reading."""
import os

import numpy as np
import pytest

import layout_utils as lu
from oracle import layout_oracle as lo
from phasm_amd.io import gfa
from phasm_amd.overlapper import ExactOverlapper

CASES = lu.load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_sequential_oracle_matches_reference_classes(case):
    exp = case["expect"]
    L = lu.node_lengths(case["lengths"])
    got = lo.layout_sequential(case["rows"], L, **case["params"])
    ovl = [lo.overlap_length(*r[2:6]) for r in case["rows"].tolist()]
    hang = [lo.overhang(r[2], r[3], r[4], r[5], L[r[0]], L[r[1]]) for r in case["rows"].tolist()]
    if case["digests"]:
        assert [got["types"].count(t) for t in range(4)] == exp["type_hist"]
        assert lu.digest(got["types"]) == exp["types_sha256"]
        assert lu.digest(ovl) == exp["overlap_len_sha256"]
        assert lu.digest(hang) == exp["overhang_sha256"]
        assert len(got["passed"]) == exp["n_passed"] and lu.digest(got["passed"]) == exp["passed_sha256"]
    else:
        assert got["types"] == exp["types"]
        assert ovl == exp["overlap_len"]
        assert [int(x) for x in hang] == exp["overhang"]
        assert got["passed"] == exp["passed"]
    assert len(case["names"]) == exp["n_segments"]
    assert [f["name"] for f in got["filters"]] == [f["name"] for f in exp["filters"]]
    for g, e in zip(got["filters"], exp["filters"]):
        assert g["filtered"] == e["filtered"], g["name"]
        assert sorted(lu.node_name(case["names"], n) for n in g["nodes_to_remove"]) == e["nodes_to_remove"], g["name"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_vectorised_form_equals_sequential(case):
    L = lu.node_lengths(case["lengths"])
    seq = lo.layout_sequential(case["rows"], L, **case["params"])
    vec = lo.layout_vectorised(case["rows"], L, **case["params"])
    assert vec["types"].tolist() == seq["types"]
    contained = np.zeros(len(case["names"]), dtype=bool)
    for n in seq["filters"][0]["nodes_to_remove"]:
        contained[n >> 1] = True
    assert vec["contained"].tolist() == contained.tolist()
    assert vec["edges"].tolist() == lo.edges_dict_to_array(seq["edges"]).tolist()


def test_final_edge_keys_do_not_depend_on_line_order():
    rng = np.random.default_rng(5)
    for case in CASES[:30]:
        L = lu.node_lengths(case["lengths"])
        base = set(lo.layout_sequential(case["rows"], L, **case["params"])["edges"])
        for _ in range(3):
            perm = rng.permutation(len(case["rows"]))
            assert set(lo.layout_sequential(case["rows"][perm], L, **case["params"])["edges"]) == base


# ---- native GFA2 reader: host-side logic of the C ABI, runs without a GPU ---------------------------

@pytest.mark.parametrize("case", CASES[:6] + CASES[11:31], ids=[c["name"] for c in CASES[:6] + CASES[11:31]])
def test_native_gfa_reader_matches_python_reading(case, tmp_path):
    p = tmp_path / "in.gfa"
    p.write_text(case["text"])
    ov = ExactOverlapper()
    nseg, res = ov.add_gfa(str(p))
    names, lengths, rows = gfa.read_gfa2_rows(case["text"].splitlines(True))
    assert nseg == len(names)
    assert ov.ids() == [n + s for n in names for s in "+-"]
    assert ov.lengths().tolist() == np.repeat(lengths, 2).tolist()
    got = res.rows()
    assert [list(map(int, r)) for r in got.tolist()] == rows.tolist()
    res.free()
    with pytest.raises(ValueError):
        ov.add_sequence("x+", "ACGT")          # a segment handle takes no sequences
    with pytest.raises(ValueError):
        ov.overlaps(3)
    ov.close()


def test_native_gfa_reader_edge_cases(tmp_path):
    text = ("H\tVN:z:2.0\r\n"
            "S\tr a\t12\t*\r\n"                      # CRLF, name with a blank
            "S\tq\t7\tACGTACG\n"
            "S\tr a\t15\t*\n"                        # same name again: the last length wins (dict)
            "E\t*\tq-\tr a+\t0\t7$\t3\t10\t*\tTS:i:5\n"
            "E\t*\tr a-\tq+\t 2\t5\t0\t3$\t*")         # no trailing newline, blank before a number
    p = tmp_path / "e.gfa"
    p.write_bytes(text.encode())
    ov = ExactOverlapper()
    nseg, res = ov.add_gfa(str(p))
    assert nseg == 2 and ov.ids() == ["r a+", "r a-", "q+", "q-"]
    assert ov.lengths().tolist() == [15, 15, 7, 7]
    assert [list(map(int, r)) for r in res.rows().tolist()] == [[3, 0, 0, 7, 3, 10], [1, 2, 2, 5, 0, 3]]
    names, lengths, rows = gfa.read_gfa2_rows(text.splitlines(True))
    assert names == ["r a", "q"] and lengths.tolist() == [15, 7] and rows.tolist() == [[3, 0, 0, 7, 3, 10], [1, 2, 2, 5, 0, 3]]
    ov.close()
    for bad in ("S\tx\t5\t*\nE\t*\tx+\ty+\t0\t1\t0\t1\t*\n",      # unknown segment (KeyError in the reference)
                "S\tx\t5\t*\nE\t*\tx+\tx\t0\t1\t0\t1\t*\n",       # no strand character
                "S\tx\tfive\t*\n",                                 # int() fails
                "S\tx\t5\n",                                       # no sequence field (IndexError)
                "S\tx\t5\t*\nE\t*\tx+\tx-\t0\t1\t0\n"):           # short edge line
        p.write_text(bad)
        ov = ExactOverlapper()
        with pytest.raises(ValueError):
            ov.add_gfa(str(p))
        ov.close()
    ov = ExactOverlapper()
    with pytest.raises(ValueError):
        ov.add_gfa(str(tmp_path / "missing.gfa"))
    ov.add_sequence("a+", "ACGT")
    p.write_text("S\tx\t5\t*\n")
    with pytest.raises(ValueError):
        ov.add_gfa(str(p))                       # needs an empty handle
    with pytest.raises(ValueError):
        ov.add_segment("x", 5)                   # sequences and segments do not mix
    ov.close()


def test_result_from_rows_round_trip_and_pairing_check():
    ov = ExactOverlapper()
    ov.add_segment("a", 10)
    ov.add_segment(b"b", 20)
    rows = np.array([[0, 2, 5, 10, 0, 5], [3, 1, 0, 4, 6, 10]], dtype=np.int64)
    res = ov.result_from_rows(rows)
    assert [list(map(int, r)) for r in res.rows().tolist()] == rows.tolist()
    res.free()
    ov.close()


def test_native_gfa_reader_survives_mangled_files(tmp_path):
    """Host-side robustness: a few hundred random mutations of a valid file (bytes flipped, fields dropped,
    lines cut, tabs and dollars sprinkled) must give either the same answer as the Python reading or a clean
    ValueError -- never a crash and never rows that name a node the handle does not hold."""
    import random
    rng = random.Random(99)
    base = CASES[12]["text"] + CASES[len(CASES) - 1]["text"].replace("H\tVN:z:2.0\tTS:i:100\n", "")
    p = tmp_path / "m.gfa"
    ok = bad = 0
    for trial in range(300):
        data = bytearray(base.encode())
        for _ in range(rng.randint(1, 2)):
            kind = rng.random()
            pos = rng.randrange(len(data))
            if kind < 0.3:
                data[pos] = rng.choice(b"\t\n$+-0123456789ESx *")
            elif kind < 0.5:
                del data[pos:pos + rng.randint(1, 12)]
            elif kind < 0.7:
                data[pos:pos] = bytes(rng.choice(b"\t\n$+-9E") for _ in range(rng.randint(1, 4)))
            elif kind < 0.85:
                cut = data.find(b"\n", pos)
                if cut > 0:
                    del data[pos:cut]
            else:
                data[pos:pos] = b"E\t*\tnope+\tnope-\t1\t2\t3\t4\t*\n" if rng.random() < 0.5 else b"S\tdup\t7\t*\nS\tdup\t9\t*\n"
        p.write_bytes(bytes(data))
        ov = ExactOverlapper()
        try:
            nseg, res = ov.add_gfa(str(p))
        except ValueError:
            bad += 1
            ov.close()
            continue
        rows = res.rows()
        n_nodes = len(ov)
        assert n_nodes == 2 * nseg
        if len(rows):
            assert int(rows["a_idx"].max()) < n_nodes and int(rows["b_idx"].max()) < n_nodes
        try:    # where the Python reading also succeeds it must agree
            text = bytes(data).decode()
            names, lengths, prows = gfa.read_gfa2_rows(text.splitlines(True))
        except Exception:
            names = None
        if names is not None and "\r" not in text and "\x0b" not in text and "\x0c" not in text:
            assert ov.ids() == [n + s for n in names for s in "+-"], trial
            assert [list(map(int, r)) for r in rows.tolist()] == prows.tolist(), trial
        res.free()
        ov.close()
        ok += 1
    assert ok > 10 and bad > 10, (ok, bad)
