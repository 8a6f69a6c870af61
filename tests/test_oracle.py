"""The oracle is pinned here: both CPU restatements must reproduce every golden vector that
was produced by the compiled reference (tests/golden/make_golden.py), and -- where the
reference binary is present (build container) -- agree with it live on fresh random cases."""
import numpy as np
import pytest

from oracle import overlap_oracle as oo
import golden_utils as gu


def test_brute_force_matches_reference_goldens():
    cases = gu.all_small_cases()
    assert len(cases) > 400
    for name, seqs, m, want in cases:
        got = oo.brute_force(seqs, m)
        assert np.array_equal(got, want), name


def test_c_oracle_matches_reference_goldens_small():
    for name, seqs, m, want in gu.all_small_cases():
        got = oo.oracle_overlaps(seqs, m)
        assert np.array_equal(got, want), name


def test_c_oracle_matches_reference_goldens_repeats():
    for name, seqs, m, want in gu.repeats_cases():
        got = oo.oracle_overlaps(seqs, m)
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("name", gu.LADDER_NAMES)
def test_c_oracle_matches_reference_goldens_ladder(name):
    _, seqs, m, want = gu.ladder_case(name)
    got = oo.oracle_overlaps(seqs, m)
    assert np.array_equal(got, want)


def test_strand_mirror_closure_on_golden():
    """with both strands added, the reference's output is closed under
    A: (a,b,la-l,la,0,l) <-> (flip b, flip a, lb-l, lb, 0, l)   and
    B: (a,b,p,p+lb,0,lb) <-> (flip a, flip b, la-p-lb, la-p, 0, lb)."""
    _, seqs, m, rows = gu.ladder_case("ladder_varlen")
    lens = np.array([len(s) for s in seqs])
    a, b, s, e, _, l = rows.T
    is_a = e == lens[a]                      # suffix of a
    is_b = l == lens[b]                      # whole of b
    # classify: rows that are both appear twice (one A, one B); mirror each multiset separately
    def mirror_a(r):
        a, b, s, e, z, l = r.T
        return np.stack([b ^ 1, a ^ 1, lens[b] - l, lens[b], z, l], axis=1)
    def mirror_b(r):
        a, b, s, e, z, l = r.T
        return np.stack([a ^ 1, b ^ 1, lens[a] - e, lens[a] - s, z, l], axis=1)
    only_a = rows[is_a & ~is_b]
    only_b = rows[is_b & ~is_a]
    both = rows[is_a & is_b]
    # every A+B row has even multiplicity (one from each family)
    uniq, cnt = np.unique(both, axis=0, return_counts=True)
    assert (cnt % 2 == 0).all()
    # full closure: mirroring A-family and B-family rows reproduces the multiset
    fam_a = np.concatenate([only_a, uniq.repeat(cnt // 2, axis=0)])
    fam_b = np.concatenate([only_b, uniq.repeat(cnt // 2, axis=0)])
    mirrored = np.concatenate([mirror_a(fam_a), mirror_b(fam_b)])
    assert np.array_equal(oo.sort_rows(mirrored), rows)


@pytest.mark.skipif(not oo.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
def test_c_oracle_matches_live_reference_random():
    rng = np.random.default_rng(99)
    for t in range(60):
        glen = int(rng.integers(50, 400))
        genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        reads = []
        for _ in range(int(rng.integers(2, 30))):
            ln = int(rng.integers(1, min(120, glen)))
            st = int(rng.integers(0, glen - ln + 1))
            r = genome[st:st + ln]
            reads.append(r)
            if rng.random() < 0.5:
                reads.append(r.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])
        m = int(rng.integers(1, 40))
        want, _, _ = oo.reference_overlaps(reads, m)
        assert np.array_equal(oo.oracle_overlaps(reads, m), want), t


def test_extension_oracle_with_zero_differences_is_the_exact_contract():
    """oracle/extend_oracle.c (the CPU restatement of po_overlaps_ex) is PARITY UNPINNED for max_diff > 0 -- the
    reference is exact -- but with max_diff = 0 it must be the exact contract, and that is pinned here by every
    reference golden (the band is ignored without differences)."""
    from oracle import extend_oracle as eo
    for name, seqs, m, want in gu.all_small_cases() + gu.repeats_cases():
        assert np.array_equal(eo.oracle_overlaps_ex(seqs, m, 0, 5), want), name
    for name in ("ladder_small", "ladder_varlen"):
        _, seqs, m, want = gu.ladder_case(name)
        assert np.array_equal(eo.oracle_overlaps_ex(seqs, m, 0, 0), want), name
    # a hand case with one substitution and one insertion (x = a[p:], y = b)
    a = "TTTTTTTTTTACGTACGGATCAGGCATCAGCATTTACGACGGATCAGCTAC"
    b = "ACGTACGGATCAGGCATGAGCATTTACGACGGATCAGCTACGGGGGGGG"      # C -> G at offset 17 of the overlap
    assert len(eo.oracle_overlaps_ex([a, b], 20, 0, 0, 8)) == 0
    rows = eo.oracle_overlaps_ex([a, b], 20, 1, 1, 8)
    assert rows.tolist() == [[0, 1, 10, len(a), 0, len(a) - 10]]
    b2 = b[:25] + "T" + b[25:]                                        # plus one inserted base in b
    assert len(eo.oracle_overlaps_ex([a, b2], 20, 1, 2, 8)) == 0
    rows = eo.oracle_overlaps_ex([a, b2], 20, 2, 2, 8)
    assert rows.tolist() == [[0, 1, 10, len(a), 0, len(a) - 10 + 1]]
