"""The other producer of the overlap file's wire format: DBdump / LAdump -> GFA2.

Expected values in tests/golden/daligner_cases.json come from the reference's own parsers and line
formatter (tests/golden/make_daligner_golden.py); nothing here reads /root/reference."""
import io
import json
import os

import numpy as np
import pytest

from phasm_amd import cli
from phasm_amd.io import daligner, gfa

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "daligner_cases.json")))
CASES = GOLDEN["cases"]
EXC = {"ValueError": ValueError, "KeyError": KeyError, "IndexError": IndexError}


def _plain(rec):
    r = dict(rec)
    if "strand" in r:
        r["strand"] = int(r["strand"])
    for k in ("arange", "brange"):
        if k in r:
            r[k] = list(r[k])
    if "trace_points" in r:
        r["trace_points"] = [list(t) for t in r["trace_points"]]
    return r


def _check(expect, fn):
    if "raises" in expect:
        with pytest.raises(EXC[expect["raises"]]):
            fn()
        return None
    got = fn()
    assert got == expect["ok"]
    return got


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_parsers_match_reference_records(case):
    _check(case["reads"], lambda: [_plain(r) for r in daligner.parse_reads(io.StringIO(case["db"]))])
    _check(case["alignments"], lambda: [_plain(r) for r in daligner.parse_local_alignments(io.StringIO(case["las"]))])


def _convert(case):
    out = io.StringIO()
    daligner.write_gfa(out, io.StringIO(case["db"]), io.StringIO(case["las"]), case["with_sequences"],
                       case["spacing"], case["translations"])
    return out.getvalue()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gfa_text_matches_reference(case):
    _check(case["gfa"], lambda: _convert(case))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_rows_equal_reading_the_converted_file(case):
    """to_rows == daligner2gfa followed by the GFA2 reader of `phasm layout`."""
    def rows():
        return daligner.to_rows(io.StringIO(case["db"]), io.StringIO(case["las"]), case["translations"])
    if "raises" in case["gfa"]:
        if case["name"] == "short_trace_line":      # only the trace-point column of the text form fails there
            rows()
            return
        with pytest.raises(EXC[case["gfa"]["raises"]]):
            rows()
        return
    names, lengths, r = rows()
    if case["name"] == "trace_empty_list":
        # `T 1` without trace lines and -t given: the converter writes an EMPTY last field, which the layout
        # reader's strip() + parts[8] then rejects (phasm/io/gfa.py:79-81) -- in the reference as well
        with pytest.raises(IndexError):
            gfa.read_gfa2_rows(io.StringIO(case["gfa"]["ok"]))
        assert r.tolist() == [[0, 3, 4, 8, 0, 4]]
        return
    n2, l2, r2 = gfa.read_gfa2_rows(io.StringIO(case["gfa"]["ok"]))
    assert names == n2
    assert np.array_equal(lengths, l2)
    assert np.array_equal(r, r2)


def test_moviename_hash_and_pacbio_names():
    for name, h in GOLDEN["moviename_hash"].items():
        assert daligner.generate_moviename_hash(name) == h
    fx = GOLDEN["fix_header"]
    for i, (seq, name) in enumerate(fx["out"]):
        assert daligner.pacbio_name(fx["moviename"], i, len(seq)) == name
        assert name in fx["map"]


def test_command(tmp_path):
    case = next(c for c in CASES if c["name"] == "random_1")      # has a translation map with descriptions
    (tmp_path / "db.txt").write_text(case["db"])
    (tmp_path / "las.txt").write_text(case["las"])
    (tmp_path / "t.json").write_text(json.dumps(case["translations"]))
    out = tmp_path / "o.gfa"
    assert cli.main(["daligner2gfa", "-T", str(tmp_path / "t.json"), "-o", str(out),
                     str(tmp_path / "db.txt"), str(tmp_path / "las.txt")]) == 0
    assert out.read_text() == case["gfa"]["ok"]
    case = next(c for c in CASES if c["name"] == "random_0")      # sequences + trace points
    (tmp_path / "db.txt").write_text(case["db"])
    (tmp_path / "las.txt").write_text(case["las"])
    assert cli.main(["daligner2gfa", "-s", "-t", "100", "-o", str(out), "-T", str(tmp_path / "absent.json"),
                     str(tmp_path / "db.txt"), str(tmp_path / "las.txt")]) == 0
    assert out.read_text() == case["gfa"]["ok"]
