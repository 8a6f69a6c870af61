"""The streamed step of po_overlaps_to_host (reads uploaded piece by piece under the kernels, reversed strand-mirror
order, deferred containments): the same rows as the goldens / the oracle for every cut of the read set into pieces."""
import os

import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo   # row helpers only
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu

CUTS = ["", "500", "250,500,750", "100,200,300,400,500,600,700,800,900", "30,60,90,120,150,180,210,240,270,300,900,950,990", "999",
        ",".join(str(c) for c in range(40, 1000, 40))]   # (24 cuts: the library keeps the first 15 = 16 pieces)


def streamed_rows(seqs, m, calls=1):
    guard = ck.GuardedReads(seqs)
    ov = ExactOverlapper()
    guard.add_all(ov)
    out = []
    for _ in range(calls):
        ov.invalidate()
        res = ov.overlaps_to_host_result(m)
        out.append((oo.sort_rows(oo.struct_to_rows(res.rows_view())), ov.stats()))
        res.free()
    ov.close()
    guard.verify_and_close()
    return out


@pytest.mark.parametrize("index", ["narrow", "wide"])
@pytest.mark.parametrize("cuts", CUTS)
def test_streamed_step_matches_the_goldens(cuts, index, monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_INDEX", index)   # (wide needs min_length >= 63: the other cases stay narrow)
    if cuts:
        monkeypatch.setenv("PHASM_STREAM_CUTS", cuts)
    n_streamed = 0
    n_wide = 0
    cases = [gu.ladder_case(name) for name in gu.LADDER_NAMES] + list(gu.repeats_cases()) + list(gu.all_small_cases())
    for name, seqs, m, want in cases:
        for got, st in streamed_rows(seqs, m, calls=2):
            ck.assert_same_rows(got, want, seqs, m, "%s, cuts %r" % (name, cuts))
            n_streamed += st["streamed"]
            n_wide += st["streamed"] and st["wide_index"]
            if st["streamed"]:
                assert st["paired"] == 1 and st["n_rows"] == len(want)
    assert n_streamed >= 2 * len(gu.LADDER_NAMES)   # every ladder case is a set of strand pairs of pure ACGT reads
    assert (n_wide >= 6) == (index == "wide")


@pytest.mark.parametrize("mode", ["1", "2"])
def test_two_stream_pieces_give_the_same_rows(mode, monkeypatch):
    """PHASM_TWO_STREAM: the counting pass of piece k + 1 beside the second half of piece k -- 1: on a stream of its own,
    2: on the stream that writes the reverse complements, held back until piece k's verify kernel starts.
    Switches, not the default; the rows must not know.  Three calls: the third runs on predicted candidate counts."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "200,400,600,800")
    monkeypatch.setenv("PHASM_TWO_STREAM", mode)
    monkeypatch.setenv("PHASM_VERIFY_ORDER", "1")
    for name in ("ladder_cfg2_mini", "cfg2_1k", "ladder_varlen"):
        _, seqs, m, want = gu.ladder_case(name)
        for got, st in streamed_rows(seqs, m, calls=3):
            assert st["streamed"] == 1
            ck.assert_same_rows(got, want, seqs, m, "%s, two-stream mode %s" % (name, mode))
    for name, seqs, m, want in gu.repeats_cases():
        for got, st in streamed_rows(seqs, m, calls=2):
            ck.assert_same_rows(got, want, seqs, m, "%s, two-stream mode %s" % (name, mode))


def _nested_reads(seed, n_reads, genome_len, lo, hi):
    """reads of very different lengths from a short genome: many reads lie inside longer ones (containments), in
    both index directions; every read is followed by its reverse complement"""
    rng = np.random.default_rng(seed)
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=genome_len))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for _ in range(n_reads):
        ln = int(rng.integers(lo, hi))
        s = int(rng.integers(0, genome_len - ln))
        r = genome[s:s + ln]
        seqs += [r, r.translate(comp)[::-1]]
    return seqs


@pytest.mark.parametrize("index,m", [("narrow", 35), ("wide", 70)])
@pytest.mark.parametrize("cuts", ["", "500", "200,400,600,800", "999"])
def test_containments_of_reads_that_arrive_later_are_deferred(cuts, index, m, monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_INDEX", index)
    if cuts:
        monkeypatch.setenv("PHASM_STREAM_CUTS", cuts)
    seqs = _nested_reads(20260, 150, 6000, 40, 1500)
    want = ck.oracle_overlaps(seqs, m)
    (got, st), (got2, st2) = streamed_rows(seqs, m, calls=2)
    assert st["streamed"] == 1 and st2["streamed"] == 1 and st["wide_index"] == (index == "wide")
    if cuts != "999":
        assert st["n_deferred"] > 0
    ck.assert_same_rows(got, want, seqs, m, "nested reads, cuts %r" % cuts)
    ck.assert_same_rows(got2, want, seqs, m, "nested reads, second call, cuts %r" % cuts)


def test_deferred_list_overflow_falls_back_to_the_chunked_form(monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "250,500,750")
    monkeypatch.setenv("PHASM_DEFER_CAP", "3")
    seqs = _nested_reads(77, 120, 5000, 40, 1200)
    m = 35
    want = ck.oracle_overlaps(seqs, m)
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    res = ov.overlaps_to_host_result(m)
    st = ov.stats()
    assert st["streamed"] == 0 and st["n_deferred"] > 3   # the list overflowed: nothing of the streamed attempt was handed out
    ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(res.rows_view())), want, seqs, m, "overflow fallback")
    res.free()
    monkeypatch.delenv("PHASM_DEFER_CAP")
    ov.invalidate()
    res = ov.overlaps_to_host_result(m)   # the next streamed call sizes the list by what the first one counted
    st = ov.stats()
    assert st["streamed"] == 1 and st["n_deferred"] > 3
    ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(res.rows_view())), want, seqs, m, "after the overflow")
    res.free()
    ov.close()


def test_streamed_fuzz_against_the_oracle(monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")
    rng = np.random.default_rng(4711)
    for trial in range(int(os.environ.get("PHASM_SOAK_TRIALS", "12"))):   # (a soak run sets a few hundred)
        cuts = sorted(set(int(c) for c in rng.integers(1, 999, size=int(rng.integers(1, 9)))))
        monkeypatch.setenv("PHASM_STREAM_CUTS", ",".join(str(c) for c in cuts))
        n = int(rng.integers(4, 90))
        glen = int(rng.integers(300, 4000))
        seqs = _nested_reads(1000 + trial, n, glen, 20, max(40, glen // 3))
        if trial % 3 == 0:   # a few exact duplicates and a period-3 repeat read
            seqs += seqs[:4] + [b"ACG" * 60, b"CGT" * 60]
        m = int(rng.integers(1, 60)) if trial % 2 else int(rng.integers(63, 120))
        monkeypatch.setenv("PHASM_INDEX", "narrow" if trial % 2 else "wide")
        want = ck.oracle_overlaps(seqs, m)
        (got, st), = streamed_rows(seqs, m)
        # (tiny min_length: more containment candidates of later reads than the deferred list holds -> the chunked form)
        assert st["streamed"] == 1 or st["n_deferred"] > 65536 or not any(len(x) >= m for x in seqs)
        ck.assert_same_rows(got, want, seqs, m, "fuzz trial %d, cuts %s, m %d" % (trial, cuts, m))


def _with_exceptions(seqs, rng, n_hits):
    """put non-ACGT bytes into strand pairs the way a FASTA with N / IUPAC / soft-masked bases gives them: byte c at
    position i of x, its complement at the mirrored position of the reverse complement"""
    comp = {b"N": b"N", b"n": b"n", b"R": b"Y", b"Y": b"R", b"a": b"t", b"c": b"g", b"g": b"c", b"t": b"a", b"W": b"W"}
    keys = list(comp)
    seqs = [bytearray(s) for s in seqs]
    for _ in range(n_hits):
        pair = int(rng.integers(0, len(seqs) // 2))
        L = len(seqs[2 * pair])
        # anywhere, with a bias to the ends and the anchor region (first / last 32 bases)
        i = int(rng.choice([0, 1, 31, 32, L - 1, L - 2, L - 32, int(rng.integers(0, L))]))
        c = keys[int(rng.integers(len(keys)))]
        seqs[2 * pair][i] = c[0]
        seqs[2 * pair + 1][L - 1 - i] = comp[c][0]
    return [bytes(s) for s in seqs]


@pytest.mark.parametrize("cuts", ["", "300,700", "100,200,300,400,500,600,700,800,900"])
def test_reads_with_exception_records_stream_too(cuts, monkeypatch):
    """N / IUPAC / lower-case bytes in a read and, mirrored, in its reverse complement (the reference compares raw
    bytes, /root/reference/src/overlapper.h:26; assembler.py:37 passes sequences as they are): the reads are kept as
    2-bit codes + sparse exception records, only the even reads travel, the odd ones are rebuilt on the device with
    the records' positions cleared, every verify -- the deferred containments' too -- consults the records."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_VERIFY_GENERATED", "1")    # the device's odd store must equal the host's, word for word
    if cuts:
        monkeypatch.setenv("PHASM_STREAM_CUTS", cuts)
    rng = np.random.default_rng(61)
    for trial in range(6):
        base = _nested_reads(100 + trial, 60, 5000, 60, 1200)
        seqs = _with_exceptions(base, rng, n_hits=int(rng.integers(1, 25)))
        m = int(rng.choice([33, 40, 64]))
        for (got, st) in streamed_rows(seqs, m, calls=2):
            assert st["streamed"] == 1 and st["paired"] == 1 and st["bits_per_base"] == 2, st
            ck.assert_same_rows(got, ck.oracle_overlaps(seqs, m), seqs, m, "exception records, trial %d, cuts %r" % (trial, cuts))
        # the unstreamed forms of the same handle state: half upload + device rebuild as well
        monkeypatch.setenv("PHASM_STREAM", "0")
        (got, st), = streamed_rows(seqs, m)
        monkeypatch.setenv("PHASM_STREAM", "1")
        assert st["streamed"] == 0 and st["paired"] == 1 and st["upload_bytes"] < sum(len(x) for x in seqs) // 4 * 0.75
        ck.assert_same_rows(got, ck.oracle_overlaps(seqs, m), seqs, m, "exception records, unstreamed")


def test_streamed_step_is_not_taken_for_reads_it_cannot_serve(monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")
    rng = np.random.default_rng(5)
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=3000))
    plain = [genome[int(s):int(s) + 400] for s in rng.integers(0, 2600, size=20)]   # not strand pairs
    broken = _nested_reads(9, 10, 2000, 100, 500)
    broken[4] = broken[4][:50] + b"N" + broken[4][51:]                               # N on one strand only: no longer a pair
    dense = [bytes(b"ACGTNRYKM"[i] for i in rng.integers(0, 9, size=300)) for _ in range(6)]
    comp = bytes.maketrans(b"ACGTNRYKM", b"TGCANYRMK")
    dense = [x for r in dense for x in (r, r.translate(comp)[::-1])]                 # strand pairs, but 8 bits per base
    for seqs, bits in ((plain, 2), (broken, 2), (dense, 8)):
        (got, st), = streamed_rows(seqs, 30)
        assert st["streamed"] == 0 and st["bits_per_base"] == bits
        ck.assert_same_rows(got, ck.oracle_overlaps(seqs, 30), seqs, 30, "not streamed")


def test_other_entry_points_after_a_streamed_step(monkeypatch):
    """The streamed step leaves the handle as po_upload would: the resident call, the chunked host call, the sharded
    calls and the extension mode on the same handle afterwards give the same rows (index built by piece 0 reused)."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "200,500,800")
    for name in ("ladder_varlen", "cfg2_1k"):
        _, seqs, m, want = gu.ladder_case(name)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        res = ov.overlaps_to_host_result(m)
        assert ov.stats()["streamed"] == 1
        ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(res.rows_view())), want, seqs, m, "%s streamed" % name)
        res.free()
        res = ov.overlaps_result(m)
        st = ov.stats()
        ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(res.rows())), want, seqs, m, "%s resident after streamed" % name)
        assert st["streamed"] == 0
        assert st["index_reused"] == (0 if os.environ.get("PHASM_POISON") else 1)   # (the poison mode rebuilds per call)
        res.free()
        res = ov.overlaps_to_host_result(m)      # not invalidated: the chunked form
        assert ov.stats()["streamed"] == 0
        ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(res.rows_view())), want, seqs, m, "%s chunked after streamed" % name)
        res.free()
        parts = np.concatenate([ov.overlaps_shard_array(m, k, 3) for k in range(3)])
        ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(parts)), want, seqs, m, "%s shards after streamed" % name)
        ck.assert_same_rows(oo.sort_rows(oo.struct_to_rows(ov.overlaps_ex_array(m, 0, 0))), want, seqs, m, "%s DP after streamed" % name)
        ov.close()


def test_streamed_step_with_empty_and_tiny_reads(monkeypatch):
    """Empty reads and reads shorter than a packed word, in the middle of the set and as its last pair: their footprint
    in the packed store is one or two words, which is what the first-words pass writes ahead of the pieces."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "300,600,900")
    base = _nested_reads(3, 30, 1500, 40, 400)
    tiny = [b"", b"", b"ACG", b"CGT", b"A", b"T"]
    # (wide, 160: min_length >= 5 W - 1, so FIVE words per read travel ahead of the pieces and the index holds windows of 4 --
    # k_scatter_lead must stop at the words a short read owns: the reads of 33 .. 130 bases own two to five of them)
    mid = [b"ACGTTGCA" * 5 + b"G", (b"ACGTTGCA" * 5 + b"G").translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]]
    for index, m in (("narrow", 1), ("narrow", 30), ("wide", 64), ("wide", 160)):
        monkeypatch.setenv("PHASM_INDEX", index)
        for seqs in (base[:20] + tiny + base[20:], base + tiny, tiny[:2] + base + tiny[:2], base[:10] + mid + tiny + base[10:] + mid):
            want = ck.oracle_overlaps(seqs, m)
            (got, st), (got2, _) = streamed_rows(seqs, m, calls=2)
            assert st["streamed"] == 1
            ck.assert_same_rows(got, want, seqs, m, "tiny reads, %s index, m %d" % (index, m))
            ck.assert_same_rows(got2, want, seqs, m, "tiny reads, second call")
