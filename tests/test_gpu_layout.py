"""Stage 1 of `phasm layout` on the device (po_layout_edges) against the layout oracle, whose filter
decisions are pinned by the reference's own classes (tests/golden/layout_cases.json).  Bit-exact
integers; the edge list is compared as a sorted set (the reference's graph has no edge order)."""
import os

import numpy as np
import pytest

import golden_utils as gu
import layout_utils as lu
import checker as ck   # the layout restatement runs in the checker process (tests/checker.py)
from phasm_amd import layout, synth
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu

CASES = lu.load_cases()


def edge_array(e):
    arr = np.stack([e["u"], e["v"], e["weight"], e["overlap_len"]], 1).astype(np.int64) if len(e) else np.empty((0, 4), np.int64)
    return arr[np.lexsort((arr[:, 1], arr[:, 0]))] if len(arr) else arr


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_layout_edges_from_gfa_file_match_oracle(case, tmp_path):
    p = tmp_path / "in.gfa"
    p.write_text(case["text"])
    got = layout.layout_from_gfa(str(p), **case["params"])
    L = lu.node_lengths(case["lengths"])
    want = ck.layout_sequential(case["rows"], L, **case["params"])
    assert np.array_equal(edge_array(got.edges), want["edges_array"])
    contained = np.zeros(len(case["names"]), dtype=bool)
    for n in want["filters"][0]["nodes_to_remove"]:
        contained[n >> 1] = True
    assert got.contained.tolist() == contained.tolist()
    st = got.stats
    assert st["n_rows"] == len(case["rows"])
    assert st["n_type"] == [want["types"].count(t) for t in range(4)]       # = the reference's classify() counts
    assert st["n_contained_reads"] == int(contained.sum())
    assert st["n_edges"] == len(want["edges"])
    # the sequential pass count is a lower bound of the per-row predicate count (it also drops rows whose
    # read was already known to be contained)
    assert st["n_pass"] >= len(want["passed"])
    assert st["n_pass"] + st["n_short"] + st["n_min_overlap"] + st["n_overhang"] == st["n_type"][0] + st["n_type"][1]


@pytest.mark.parametrize("name", ["ladder_varlen", "ladder_cfg1_mini", "cfg1_full"])
def test_overlap_then_layout_without_a_file(name):
    """Rows straight from po_overlaps (still in HBM) into po_layout_edges."""
    _, seqs, m, _ = gu.ladder_case(name)
    ov = ExactOverlapper()
    for i in range(len(seqs) // 2):
        ov.add_sequence("read%d+" % i, seqs[2 * i])
        ov.add_sequence("read%d-" % i, seqs[2 * i + 1])
    res = ov.overlaps_result(m)
    rows = res.rows()
    for params in (layout.DEFAULTS, dict(layout.DEFAULTS, min_read_length=5000, min_overlap_length=2000)):
        got = layout.build_assembly_graph(ov, res, **params)
        r6 = np.stack([rows[k] for k in rows.dtype.names], 1).astype(np.int64)
        want = ck.layout_vectorised(r6, ov.lengths(), **params)
        assert np.array_equal(edge_array(got.edges), want["edges"])
        assert got.contained.tolist() == want["contained"].tolist()
        # exact overlaps have no overhang and every B row is a containment
        assert got.stats["n_overhang"] == 0
        assert got.stats["n_type"][1] == 0
    res.free()
    ov.close()


def test_random_bulk_rows_against_vectorised_oracle():
    rng = np.random.default_rng(42)
    n_names = 3000
    lengths = rng.integers(200, 3000, n_names)
    L = np.repeat(lengths, 2)
    n = 400_000
    a = rng.integers(0, 2 * n_names, n)
    b = rng.integers(0, 2 * n_names, n)
    # the last 5 % of the rows: arbitrary ranges (every type, many containments) among the first 600 reads only,
    # so that the other reads stay in the graph
    g = n - n // 20
    a[g:] = rng.integers(0, 1200, n - g)
    b[g:] = rng.integers(0, 1200, n - g)
    la, lb = L[a], L[b]
    s = (rng.random(n) * la * 0.9).astype(np.int64)
    e = s + 1 + (rng.random(n) * (la - s - 1)).astype(np.int64)
    bs = (rng.random(n) * lb * 0.3).astype(np.int64)
    be = bs + 1 + (rng.random(n) * (lb - bs - 1)).astype(np.int64)
    # the rest: dovetails (a suffix of a = a prefix of b, shorter than both), some with small overhangs
    l = np.maximum(np.minimum(la[:g], lb[:g]) // 2 - rng.integers(0, 50, g), 1)
    oa = rng.integers(0, 3, g) * rng.integers(0, 20, g)
    ob = rng.integers(0, 3, g) * rng.integers(0, 20, g)
    oa = np.minimum(oa, la[:g] - l)
    ob = np.minimum(ob, lb[:g] - l)
    s[:g], e[:g], bs[:g], be[:g] = la[:g] - l - oa, la[:g] - oa, ob, ob + l
    rows = np.stack([a, b, s, e, bs, be], 1)
    ov = ExactOverlapper()
    for i in range(n_names):
        ov.add_segment("r%d" % i, int(lengths[i]))
    res = ov.result_from_rows(rows)
    for params in (layout.DEFAULTS, dict(min_read_length=400, min_overlap_length=150, max_overhang_abs=60, max_overhang_rel=0.3)):
        got = layout.build_assembly_graph(ov, res, **params)
        want = ck.layout_vectorised(rows, L, **params)
        assert np.array_equal(edge_array(got.edges), want["edges"])
        assert got.contained.tolist() == want["contained"].tolist()
        assert got.stats["n_edges"] == len(want["edges"]) > 1000
    res.free()
    ov.close()


def test_layout_needs_strand_paired_ids_and_own_rows():
    ov = ExactOverlapper()
    ov.add_sequence("a", "ACGTACGTAA")
    ov.add_sequence("b", "ACGTAAGGTT")
    res = ov.overlaps_result(3)
    with pytest.raises(ValueError):
        ov.layout_edges(res)
    other = ExactOverlapper()
    other.add_segment("x", 10)
    with pytest.raises(ValueError):
        other.layout_edges(res)
    bad = other.result_from_rows(np.array([[0, 7, 0, 5, 0, 5]]))
    with pytest.raises(ValueError):
        other.layout_edges(bad)                 # node 7 does not exist
    empty = other.result_from_rows(np.empty((0, 6), dtype=np.int64))
    e, removed = other.layout_edges(empty)
    assert len(e) == 0 and removed.tolist() == [0]
    res.free()
    ov.close()
    other.close()


def test_cli_overlap_then_layout_edges_files(tmp_path):
    """reads.fasta -> `overlap` -> overlaps.gfa -> `layout-edges` -> graph.gfa in gfa2_write_graph's
    format (phasm/io/gfa.py:283-327); the edges equal the oracle's for the rows of the overlap file."""
    from phasm_amd import cli
    from phasm_amd.io import gfa
    cfg = synth.SynthConfig(n_reads=150, read_len=5000, genome_len=40_000, ploidy=2, snp=0.005, seed=21,
                            len_sd=1500.0, len_min=1000, len_max=9000)      # variable lengths: containments
    reads = synth.generate_reads(cfg)
    fa = tmp_path / "reads.fasta"
    synth.write_fasta(str(fa), reads, width=80)
    ovl = tmp_path / "overlaps.gfa"
    graph = tmp_path / "graph.gfa"
    assert cli.main(["overlap", str(fa), "-l", "500", "-o", str(ovl)]) == 0
    assert cli.main(["layout-edges", str(ovl), "-l", "3000", "-s", "800", "-o", str(graph)]) == 0
    names, lengths, rows = gfa.read_gfa2_rows(ovl.read_text().splitlines(True))
    L = np.repeat(lengths, 2)
    want = ck.layout_sequential(rows, L, min_read_length=3000, min_overlap_length=800)
    assert len(want["edges"]) > 50
    lines = graph.read_text().splitlines(True)
    assert lines[0] == "H\tVN:z:2.0\n"
    s_lines = [l for l in lines if l.startswith("S\t")]
    e_lines = [l for l in lines if l.startswith("E\t")]
    assert len(lines) == 1 + len(s_lines) + len(e_lines)
    node = lambda n: names[n >> 1] + "+-"[n & 1]
    want_e = sorted(gfa.gfa_line("E", "*", node(u), node(v), w, int(L[u]), 0, o, "*") for (u, v), (w, o) in want["edges"].items())
    assert sorted(e_lines) == want_e
    used = sorted({n >> 1 for uv in want["edges"] for n in uv})
    assert s_lines == [gfa.gfa_line("S", names[i], int(lengths[i]), "*") for i in used]


def _daligner_cases():
    import json
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "daligner_cases.json")))
    return [c for c in d["cases"] if "ok" in c["gfa"] and c["name"] not in ("no_reads", "trace_empty_list")]


@pytest.mark.parametrize("case", _daligner_cases(), ids=[c["name"] for c in _daligner_cases()])
def test_layout_from_daligner_dumps(case, tmp_path):
    """DBdump + LAdump straight to the device == the reference-formatted GFA2 text read natively == the oracle."""
    import io
    from phasm_amd import cli
    from phasm_amd.io import gfa
    params = dict(layout.DEFAULTS, max_overhang_abs=20, max_overhang_rel=0.3)
    direct = layout.layout_from_daligner(io.StringIO(case["db"]), io.StringIO(case["las"]), case["translations"], **params)
    p = tmp_path / "in.gfa"
    p.write_text(case["gfa"]["ok"])
    via_file = layout.layout_from_gfa(str(p), **params)
    assert direct.ids == via_file.ids
    assert np.array_equal(edge_array(direct.edges), edge_array(via_file.edges))
    assert direct.contained.tolist() == via_file.contained.tolist()
    names, lengths, rows = gfa.read_gfa2_rows(io.StringIO(case["gfa"]["ok"]))
    want = ck.layout_sequential([tuple(r) for r in rows.tolist()], lu.node_lengths(lengths.tolist()), **params)
    assert np.array_equal(edge_array(direct.edges), want["edges_array"])
    # and as a command
    (tmp_path / "db.txt").write_text(case["db"])
    (tmp_path / "las.txt").write_text(case["las"])
    args = ["layout-edges", "-a", "20", "-r", "0.3", str(tmp_path / "db.txt"), "--las", str(tmp_path / "las.txt")]
    if case["translations"]:
        import json
        (tmp_path / "t.json").write_text(json.dumps(case["translations"]))
        args += ["-T", str(tmp_path / "t.json")]
    assert cli.main(args + ["-o", str(tmp_path / "g1.gfa")]) == 0
    assert cli.main(["layout-edges", "-a", "20", "-r", "0.3", str(p), "-o", str(tmp_path / "g2.gfa")]) == 0
    assert (tmp_path / "g1.gfa").read_bytes() == (tmp_path / "g2.gfa").read_bytes()


@pytest.mark.parametrize("name", ["ladder_varlen", "cfg2_1k", "cfg3_1k"])
def test_layout_without_the_table_equals_layout_with_it(name, monkeypatch):
    """Rows straight from the paired-strand emission need no dedupe table (the last row of an adjacent
    (row, strand mirror) group owns its twin edge pair): the table-free pass, the table pass
    (PHASM_LAYOUT_TABLE=1) and the oracle agree, whole-set, pipelined-to-host and repeat-rich."""
    _, seqs, m, _ = gu.ladder_case(name)
    ov = ExactOverlapper()
    for i in range(len(seqs) // 2):
        ov.add_sequence("read%d+" % i, seqs[2 * i])
        ov.add_sequence("read%d-" % i, seqs[2 * i + 1])
    outs = []
    for mode in ("fast", "table", "host", "streamed"):
        if mode == "table":
            monkeypatch.setenv("PHASM_LAYOUT_TABLE", "1")
        else:
            monkeypatch.delenv("PHASM_LAYOUT_TABLE", raising=False)
        if mode == "streamed":   # rows of the streamed step: another member of each mirror pair computed, deferred rows last
            monkeypatch.setenv("PHASM_STREAM", "1")
            monkeypatch.setenv("PHASM_STREAM_CUTS", "300,600,900")
            ov.invalidate()
        res = ov.overlaps_to_host_result(m) if mode in ("host", "streamed") else ov.overlaps_result(m)
        assert ov.stats()["streamed"] == (mode == "streamed")
        g = layout.build_assembly_graph(ov, res, min_overlap_length=m + 50)
        rows = res.rows()
        res.free()
        outs.append((edge_array(g.edges), g.contained.tolist()))
    r6 = np.stack([rows[k] for k in rows.dtype.names], 1).astype(np.int64)
    want = ck.layout_vectorised(r6, ov.lengths(), min_overlap_length=m + 50)
    ov.close()
    for e, c in outs:
        assert np.array_equal(e, want["edges"]) and c == want["contained"].tolist()
    assert len(want["edges"]) > 100


def test_layout_on_streamed_rows_of_nested_reads(monkeypatch):
    """Reads of very different lengths: many containments, part of them settled from the streamed step's deferred list
    (their rows come last in the array).  Stage 1 on those rows -- table-free -- equals the oracle's."""
    rng = np.random.default_rng(99)
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=7000))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ov = ExactOverlapper()
    for i in range(160):
        ln = int(rng.integers(60, 1800))
        st = int(rng.integers(0, 7000 - ln))
        r = genome[st:st + ln]
        ov.add_sequence("read%d+" % i, r)
        ov.add_sequence("read%d-" % i, r.translate(comp)[::-1])
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "200,400,600,800")
    m = 40
    res = ov.overlaps_to_host_result(m)
    assert ov.stats()["streamed"] == 1 and ov.stats()["n_deferred"] > 0
    g = layout.build_assembly_graph(ov, res, min_overlap_length=m + 20)
    rows = res.rows()
    res.free()
    r6 = np.stack([rows[k] for k in rows.dtype.names], 1).astype(np.int64)
    want = ck.layout_vectorised(r6, ov.lengths(), min_overlap_length=m + 20)
    ov.close()
    assert np.array_equal(edge_array(g.edges), want["edges"]) and g.contained.tolist() == want["contained"].tolist()
    assert len(want["contained"]) > 20
