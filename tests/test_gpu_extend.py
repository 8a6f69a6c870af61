"""po_overlaps_ex -- the banded seed-extension DP kernel (phasm_amd/csrc/extend.hip.h).

* max_diff = 0: the DP kernel in place of the packed compare must return exactly the rows of po_overlaps -- checked
  against the REFERENCE goldens (toy, adversarial incl. N / lower case, repeats, the synthetic ladders).  This half is
  pinned by the reference.
* max_diff > 0: the reference is exact (src/overlapper.cpp:28-150) and cannot check an inexact overlap -- PARITY
  UNPINNED.  The checker is the build's own CPU restatement (oracle/extend_oracle.c, a plain row-by-row banded DP,
  run in the checker process), on seeded read sets with substitutions and indels.  Bit-exact rows, sorted multisets.
"""
import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo   # row helpers only
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bits", "lanes", "wave"], autouse=True)
def dp_kernel(request, monkeypatch):
    """Every test runs through all three mappings of the DP: a lane per candidate with the band row as a bit vector
    (k_extend_bits, the default: 2-bit reads, band <= 15), a lane per candidate with the band row in registers
    (k_extend_lanes) and a wave per candidate with a lane per diagonal (k_extend_dp).  Where a lane mapping does not apply
    the library picks the wave kernel by itself, so the default selection is exercised too."""
    if request.param == "wave":
        monkeypatch.setenv("PHASM_DP_KERNEL", "wave")
    elif request.param == "lanes":
        monkeypatch.setenv("PHASM_DP_SORT", "1")     # candidates ordered by length (what large calls do), also on small inputs
        monkeypatch.setenv("PHASM_DP_KERNEL_SOFT", "lanes")
    else:
        monkeypatch.delenv("PHASM_DP_KERNEL", raising=False)
        monkeypatch.setenv("PHASM_DP_SORT", "1")
    return request.param


def ex_rows(seqs, m, max_diff, band):
    guard = ck.GuardedReads(seqs)
    ov = ExactOverlapper()
    guard.add_all(ov)
    arr = ov.overlaps_ex_array(m, max_diff, band)
    st = ov.stats()
    ov.close()
    guard.verify_and_close()
    return oo.sort_rows(oo.struct_to_rows(arr)), st


def test_max_diff_zero_equals_the_reference_goldens():
    for name, seqs, m, want in gu.all_small_cases() + gu.repeats_cases():
        got, st = ex_rows(seqs, m, 0, 0)
        ck.assert_same_rows(got, want, seqs, m, name)
        assert st["max_diff"] == 0


@pytest.mark.parametrize("name", ["ladder_small", "ladder_varlen", "ladder_cfg1_mini", "ladder_cfg2_mini", "ladder_cfg4_noise"])
def test_max_diff_zero_equals_the_ladder_goldens(name):
    _, seqs, m, want = gu.ladder_case(name)
    got, st = ex_rows(seqs, m, 0, 7)        # (the band is ignored without differences)
    ck.assert_same_rows(got, want, seqs, m, name)
    assert st["paired"] == 1 and st["dp_steps"] > 0
    # ... and equals the packed-compare path row for row
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    assert np.array_equal(ov.overlaps_ex_array(m, 0, 0), ov.overlaps_array(m))
    ov.close()


def noisy_reads(rng, n_reads, glen, lo, hi, sub, indel, both_strands):
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
    reads = []
    for _ in range(n_reads):
        ln = int(rng.integers(lo, hi))
        st = int(rng.integers(0, glen - ln + 1))
        r = bytearray()
        for ch in genome[st:st + ln]:
            u = rng.random()
            if u < sub:
                r.append(b"ACGT"[(b"ACGT".index(ch) + int(rng.integers(1, 4))) & 3])
            elif u < sub + indel / 2:
                continue                                   # deletion
            elif u < sub + indel:
                r.append(ch)
                r.append(b"ACGT"[int(rng.integers(4))])    # insertion
            else:
                r.append(ch)
        r = bytes(r)
        if rng.random() < 0.5:
            r = r.translate(rc)[::-1]
        reads.append(r)
    if both_strands:
        out = []
        for r in reads:
            out += [r, r.translate(rc)[::-1]]
        return out
    return reads


@pytest.mark.parametrize("seed,max_diff,band", [(1, 1, 1), (2, 3, 2), (3, 8, 4), (4, 20, 16), (5, 40, 30), (6, 5, 0), (7, 2, 30)])
def test_inexact_rows_equal_the_cpu_restatement(seed, max_diff, band, dp_kernel):
    rng = np.random.default_rng(1000 + seed)
    seqs = noisy_reads(rng, n_reads=40, glen=3000, lo=150, hi=1400, sub=0.01, indel=0.006, both_strands=seed % 2 == 0)
    m = int(rng.choice([40, 64, 100]))
    got, st = ex_rows(seqs, m, max_diff, band)
    want = ck.oracle_overlaps_ex(seqs, m, max_diff, band, anchor=32)
    assert st["paired"] == 0 and st["max_diff"] == max_diff and st["band"] == band
    assert st["dp_lanes"] == ((2 if dp_kernel == "bits" else 1) if (dp_kernel != "wave" and band <= 15) else 0)
    assert np.array_equal(got, want), (len(got), len(want), [tuple(r) for r in got[:5]], [tuple(r) for r in want[:5]])
    exact = ck.oracle_overlaps(seqs, m)
    assert len(want) >= len(exact)          # tolerance only ever adds overlaps of a pair / occurrences
    assert len(want) > len(exact)           # ... and on this noise it does


@pytest.mark.parametrize("seed,max_diff,band", [(11, 30, 8), (12, 60, 15), (13, 12, 7), (14, 25, 11), (15, 6, 3), (16, 0, 8)])
def test_long_candidates_run_through_the_blocked_rows(seed, max_diff, band, dp_kernel):
    """k_extend_bits walks a candidate in three phases (general rows until x stands at a dword boundary, blocks of 16 rows
    without a test, general rows for the end rows): reads of up to 2 600 bases starting anywhere, so that every alignment
    of x and y, every number of blocks and both kinds of end row (suffix-prefix and containment of short reads) occur, with
    indels that move the path across the band.  All three mappings against the CPU restatement."""
    rng = np.random.default_rng(2000 + seed)
    seqs = noisy_reads(rng, n_reads=56, glen=5000, lo=100, hi=2600, sub=0.008, indel=0.004, both_strands=seed % 2 == 1)
    m = int(rng.choice([64, 90, 128]))
    got, st = ex_rows(seqs, m, max_diff, band)
    want = ck.oracle_overlaps_ex(seqs, m, max_diff, band, anchor=32)
    assert st["dp_lanes"] == ((2 if dp_kernel == "bits" else 1) if dp_kernel != "wave" else 0)
    assert len(want) > (50 if max_diff else 0)       # (max_diff 0 on noisy reads: the few exact overlaps, still through the DP)
    assert np.array_equal(got, want), (len(got), len(want), [tuple(r) for r in got[:5]], [tuple(r) for r in want[:5]])
    if max_diff:
        assert st["dp_steps"] > 100 * len(want)      # (the rows were walked: long candidates, most of them in blocks)


def test_inexact_mode_on_8bit_reads_and_errors():
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    genome = alpha[rng.integers(0, len(alpha), size=4000)].tobytes()
    seqs = []
    for _ in range(40):
        ln = int(rng.integers(100, 900))
        st = int(rng.integers(0, len(genome) - ln))
        r = bytearray(genome[st:st + ln])
        for pos in rng.integers(0, ln, size=ln // 150):
            r[pos] = alpha[rng.integers(len(alpha))]
        seqs.append(bytes(r))
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    got = oo.sort_rows(oo.struct_to_rows(ov.overlaps_ex_array(30, 4, 3)))
    assert ov.stats()["bits_per_base"] == 8
    assert np.array_equal(got, ck.oracle_overlaps_ex(seqs, 30, 4, 3, anchor=8))
    with pytest.raises(ValueError):
        ov.overlaps_ex_array(30, 4, 31)      # band > 30
    ov.close()
    # sparse non-ACGT bytes on the 2-bit path: exact mode works (exception records), inexact mode refuses
    ov = ExactOverlapper()
    ov.add_sequence("a", "ACGTNACGTACGGATTACAGATTACAGGGATCCGATTTACGAGCATCGACTAGCTACGACTAGC")
    ov.add_sequence("b", "GATTACAGGGATCCGATTTACGAGCATCGACTAGCTACGACTAGCNNACGATCGATCGAAA")
    assert len(ov.overlaps_ex_array(20, 0, 0)) == len(ov.overlaps_array(20)) == 1
    with pytest.raises(ValueError):
        ov.overlaps_ex_array(20, 2, 2)
    ov.close()


def test_inexact_fuzz_small_anchors_short_reads_and_edge_bands():
    """More of the same against the CPU restatement: min_length below the word size (short anchors), reads shorter
    than the band, identical and contained reads, unpaired and odd-sized read sets, max_diff from 1 to far more than
    any read can differ by."""
    rng = np.random.default_rng(4711)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    for trial in range(24):
        glen = int(rng.integers(200, 1500))
        genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        reads = []
        for _ in range(int(rng.integers(3, 30))):
            ln = int(rng.integers(5, min(glen, 400)))
            st = int(rng.integers(0, glen - ln + 1))
            r = bytearray(genome[st:st + ln])
            for _e in range(int(rng.integers(0, 4))):
                pos = int(rng.integers(0, len(r)))
                kind = rng.integers(3)
                if kind == 0:
                    r[pos] = b"ACGT"[int(rng.integers(4))]
                elif kind == 1 and len(r) > 6:
                    del r[pos]
                else:
                    r.insert(pos, b"ACGT"[int(rng.integers(4))])
            r = bytes(r)
            reads.append(r if rng.random() < 0.5 else r.translate(rc)[::-1])
            if rng.random() < 0.15:
                reads.append(reads[-1])
        if rng.random() < 0.5:
            seqs = []
            for r in reads:
                seqs += [r, r.translate(rc)[::-1]]
        else:
            seqs = reads
        m = int(rng.choice([5, 8, 12, 31, 32, 33, 50]))
        max_diff = int(rng.choice([1, 2, 5, 30, 1000]))
        band = int(rng.choice([0, 1, 3, 9, 30]))
        got, st = ex_rows(seqs, m, max_diff, band)
        want = ck.oracle_overlaps_ex(seqs, m, max_diff, band, anchor=32)
        assert np.array_equal(got, want), (trial, m, max_diff, band, len(seqs), len(got), len(want))


def test_cli_overlap_with_max_diff_writes_the_extension_rows(tmp_path):
    """`overlap --max-diff E --band W` writes the rows of po_overlaps_ex; without the option the file is the exact one."""
    from phasm_amd import cli
    from phasm_amd.io import gfa
    rng = np.random.default_rng(2024)
    reads = noisy_reads(rng, n_reads=30, glen=2500, lo=300, hi=1200, sub=0.01, indel=0.004, both_strands=False)
    named = [("read%d" % i, r) for i, r in enumerate(reads)]
    fa = tmp_path / "reads.fasta"
    synth.write_fasta(str(fa), named, width=60)
    seqs = [s for _, s in synth.oriented(named)]
    ids = [n for n, _ in synth.oriented(named)]
    for extra, want in (([], ck.oracle_overlaps(seqs, 50)), (["--max-diff", "6", "--band", "3"], ck.oracle_overlaps_ex(seqs, 50, 6, 3))):
        out = tmp_path / ("out%d.gfa" % len(extra))
        assert cli.main(["overlap", str(fa), "-l", "50", "-o", str(out)] + extra) == 0
        e_lines = sorted(l for l in out.read_text().splitlines(keepends=True) if l.startswith("E\t"))
        want_lines = sorted(gfa.gfa_line("E", "*", ids[a], ids[b], s, e, bs, be, "*") for a, b, s, e, bs, be in want.tolist())
        assert e_lines == want_lines
        assert len(want_lines) > 10
