"""Host logic around the hot path: FASTA ingest, reverse complement, GFA2 lines, generator."""
import io

import numpy as np

from phasm_amd import synth
from phasm_amd._lib import ROW_DTYPE
from phasm_amd.io import gfa
from phasm_amd.io.fasta import read_fasta, reverse_complement


def test_gfa_lines_match_reference_format():
    # strings observed from the reference's phasm.io.gfa
    assert gfa.gfa_header() == "H\tVN:z:2.0\n"
    assert gfa.gfa_line("E", "*", "read1+", "read2+", 6, 12, 0, 6, "*") == "E\t*\tread1+\tread2+\t6\t12\t0\t6\t*\n"
    assert gfa.gfa_line("S", "read1", 12, "*") == "S\tread1\t12\t*\n"
    assert gfa.gfa_header(trace_spacing=100) == "H\tVN:z:2.0\tTS:i:100\n"


def test_write_edges_is_byte_identical_to_per_row_gfa_line():
    rows = np.array([(0, 1, 6, 12, 0, 6), (1, 0, 0, 5, 0, 5), (2, 1, 100000, 115000, 0, 15000)], dtype=ROW_DTYPE)
    ids = ["r a+", "r a-", "x|y z+"]
    want = "".join(gfa.gfa_line("E", "*", ids[r["a_idx"]], ids[r["b_idx"]], r["astart"], r["aend"],
                                r["bstart"], r["bend"], "*") for r in rows)
    out = io.StringIO()
    assert gfa.write_edges(out, rows, ids, chunk=2) == 3
    assert out.getvalue() == want


def test_fasta_reader_multiline_blank_lines_and_full_header():
    txt = b">read1 some description\nACGT\nacgt\n\n>read2\n\nNNNN\nAC\n>empty\n>last\nT\n"
    recs = list(read_fasta(io.BytesIO(txt)))
    assert recs == [("read1 some description", b"ACGTacgt"), ("read2", b"NNNNAC"), ("empty", b""), ("last", b"T")]


def test_reverse_complement():
    assert reverse_complement(b"AAACCCGGGTTT") == b"AAACCCGGGTTT"
    assert reverse_complement(b"GGGTTTACGTAC") == b"GTACGTAAACCC"
    assert reverse_complement("ACGTN") == "NACGT"
    assert reverse_complement(b"acgt") == b"acgt"
    s = synth.codes_to_ascii(np.random.default_rng(0).integers(0, 4, 1000, dtype=np.uint8))
    assert reverse_complement(reverse_complement(s)) == s


def test_generator_is_seeded_and_shaped():
    cfg = synth.SynthConfig(n_reads=50, read_len=500, genome_len=3000, ploidy=3, seed=9)
    a = synth.generate_reads(cfg)
    b = synth.generate_reads(cfg)
    assert a == b and len(a) == 50 and all(len(s) == 500 for _, s in a)
    assert a[0][0] == "read0" and set(b"".join(s for _, s in a)) <= set(b"ACGT")
    o = synth.oriented(a)
    assert len(o) == 100 and o[0][0] == "read0+" and o[1][0] == "read0-"
    assert o[1][1] == reverse_complement(o[0][1])
    c1 = synth.CONFIGS["cfg1"]
    assert c1.len_sd > 0 and synth.CONFIGS["cfg2"].n_reads == 50_000
    sc = synth.scaled(synth.CONFIGS["cfg2"], 800)
    assert sc.genome_len == 80_000 and sc.n_reads == 800


def test_native_fasta_ingest_matches_python_reader(tmp_path):
    from phasm_amd.overlapper import ExactOverlapper
    txt = (b">read1 some description\nACGT\nacgt\n\n>read2\r\n\nNNNN\r\nAC \n>empty\n>last|x y\nTTGACGTTGCAAGGT\nAC")
    path = tmp_path / "r.fasta"
    path.write_bytes(txt)
    recs = list(read_fasta(str(path)))
    ov = ExactOverlapper()
    assert ov.add_fasta(str(path)) == len(recs) == 4
    want_ids, want_len = [], []
    for name, seq in recs:
        want_ids += [name + "+", name + "-"]
        want_len += [len(seq), len(seq)]
    assert ov.ids() == want_ids
    assert ov.lengths().tolist() == want_len
    single = ExactOverlapper()
    assert single.add_fasta(str(path), both_strands=False) == 4
    assert single.ids() == [n for n, _ in recs]
    import pytest
    with pytest.raises(ValueError):
        single.add_fasta(str(tmp_path / "missing.fasta"))
    ov.close()
    single.close()


def _stores(ov):
    """Both packed host stores of a handle, as bytes (po_debug_store_words)."""
    import ctypes
    from phasm_amd import _lib
    lib = _lib.load()
    out = []
    for k in (0, 1):
        ptr = ctypes.c_void_p()
        n = lib.po_debug_store_words(ov._h, k, ctypes.byref(ptr))
        out.append(ctypes.string_at(ptr.value, n * 8) if n else b"")
    return out


def test_parallel_and_sequential_fasta_ingest_agree(tmp_path, monkeypatch):
    """po_add_fasta packs pure-ACGT files with several threads from the mapped file and falls back to the
    record-by-record path for anything else: both must give the same reads (names, lengths -- and, on the GPU,
    the same overlap file: tests/test_gpu_e2e.py)."""
    import random
    from phasm_amd.overlapper import ExactOverlapper
    rng = random.Random(3)
    recs = []
    for i in range(700):
        n = rng.choice([0, 1, 31, 32, 33, 64, 65, 700, 3000, rng.randint(1, 5000)])
        recs.append(("read %d len=%d" % (i, n), "".join(rng.choice("ACGT") for _ in range(n))))
    def write(path, recs, width, crlf=False, blanks=False):
        eol = "\r\n" if crlf else "\n"
        with open(path, "w", newline="") as f:
            f.write("junk before the first header" + eol)
            for name, seq in recs:
                f.write(">" + name + eol)
                for k in range(0, len(seq), width):
                    f.write(("  " if blanks and k == 0 else "") + seq[k:k + width] + (" \t" if blanks else "") + eol)
                if blanks:
                    f.write(eol)
    for width, crlf, blanks in ((80, False, False), (10 ** 9, False, False), (61, True, True)):
        p = tmp_path / ("r_%d_%d.fa" % (min(width, 999), crlf))
        write(str(p), recs, width, crlf, blanks)
        got = {}
        # (the record scan of the parallel path cuts the file into ranges: 1, a few, and more ranges than records per range)
        for mode in ("parallel", "ranges3", "ranges7", "ranges64", "sequential"):
            monkeypatch.delenv("PHASM_FASTA_RANGES", raising=False)
            if mode == "sequential":
                monkeypatch.setenv("PHASM_FASTA_SEQUENTIAL", "1")
            else:
                monkeypatch.delenv("PHASM_FASTA_SEQUENTIAL", raising=False)
                if mode.startswith("ranges"):
                    monkeypatch.setenv("PHASM_FASTA_RANGES", mode[6:])
            # (fresh store blocks are filled with 0xA5: the parallel path grows the stores without zero-filling them and writes
            # every word itself -- the reads' words, the alignment gaps, the spare word behind each read)
            monkeypatch.setenv("PHASM_POISON_HOST", "1")
            ov = ExactOverlapper()
            assert ov.add_fasta(str(p)) == len(recs)
            got[mode] = (ov.ids(), ov.lengths().tolist(), _stores(ov))
            ov.close()
        assert got["parallel"] == got["sequential"] == got["ranges3"] == got["ranges7"] == got["ranges64"]
        assert got["parallel"][0] == [n + s for n, _ in recs for s in "+-"]
        assert got["parallel"][1] == [len(q) for _, q in recs for _ in "+-"]
    # a file with one N: the parallel path steps aside, the result is the sequential one
    recs2 = recs[:50] + [("with N", "ACGTNACGT" * 20)] + recs[50:80]
    p = tmp_path / "n.fa"
    write(str(p), recs2, 70)
    monkeypatch.delenv("PHASM_FASTA_SEQUENTIAL", raising=False)
    ov = ExactOverlapper()
    assert ov.add_fasta(str(p)) == len(recs2)
    assert ov.ids() == [n + s for n, _ in recs2 for s in "+-"]
    assert ov.lengths().tolist() == [len(q) for _, q in recs2 for _ in "+-"]
    ov.close()


def test_native_edge_writer_many_chunks_file_and_pipe(tmp_path):
    """po_write_gfa_edges formats chunks of rows on several threads; into a file every thread writes its own bytes at its
    own offset (pwrite), into a pipe the chunks go out in order.  Both must be the bytes of the per-row reference format
    (gfa_line, /root/reference/phasm/io/gfa.py:230-231) -- here on 200 k rows, i.e. several chunks, without a GPU."""
    import os
    import threading
    from phasm_amd.overlapper import ExactOverlapper
    rng = np.random.default_rng(12)
    ov = ExactOverlapper()
    names = ["read %d|x" % i if i % 3 else "r%d" % i for i in range(60)]
    for i, n in enumerate(names):
        ov.add_segment(n, 1000 + 17 * i)
    ids = ov.ids()
    n = 200_000
    rows = np.zeros(n, dtype=ROW_DTYPE)
    rows["a_idx"] = rng.integers(0, len(ids), n)
    rows["b_idx"] = rng.integers(0, len(ids), n)
    rows["astart"] = rng.integers(0, 2_000_000_000, n)
    rows["aend"] = rng.integers(0, 20_000, n)
    rows["bstart"] = rng.integers(-5, 5, n)           # (negative numbers never occur in overlap rows; the writer takes any int32)
    rows["bend"] = rng.integers(0, 120, n)
    res = ov.result_from_rows(rows)
    py = io.StringIO()
    gfa.write_edges(py, rows, ids)
    want = py.getvalue().encode()
    path = tmp_path / "edges.gfa"
    with open(path, "w") as f:
        f.write("H\tVN:z:2.0\n")
        assert res.write_gfa_edges(f) == n
        f.write("S\tafter\t1\t*\n")                  # the descriptor's offset stands behind the edges
    assert path.read_bytes() == b"H\tVN:z:2.0\n" + want + b"S\tafter\t1\t*\n"
    rd, wr = os.pipe()
    got = []
    t = threading.Thread(target=lambda: got.append(b"".join(iter(lambda: os.read(rd, 1 << 20), b""))))
    t.start()
    with os.fdopen(wr, "w") as f:
        assert res.write_gfa_edges(f) == n
    t.join()
    os.close(rd)
    assert got[0] == want
    res.free()
    ov.close()
