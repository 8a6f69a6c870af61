"""``synth.expected_rows``: the complete A u B multiset computed from the generator's truth (haplotype, start,
length, strand of every read) without any overlapper.  It is what the FULL-SIZE GPU tests compare with
(tests/test_gpu_fullsize.py), so it is pinned here first: identical to every ladder golden, i.e. to the output of
/root/reference/src/overlapper.cpp:28-150 on the same reads (tests/golden/make_golden.py), variable lengths,
containments, three and four haplotypes and substitution noise included."""
import json
import os

import numpy as np
import pytest

import golden_utils as gu
import rowsig
from oracle import overlap_oracle as oo
from phasm_amd import synth


@pytest.mark.parametrize("name", gu.LADDER_NAMES)
def test_expected_rows_equal_the_reference_goldens(name):
    z = np.load(os.path.join(gu.GOLDEN, name + ".npz"))
    cfg = synth.SynthConfig(**json.loads(str(z["config"])))
    m = int(z["min_length"])
    want = np.asarray(z["rows"], dtype=np.int64).reshape(-1, 6)
    got = synth.expected_rows(cfg, m)
    assert np.array_equal(oo.sort_rows(got), want)
    assert rowsig.signature(got) == rowsig.signature(want)
    # chunking of the pair enumeration does not matter
    assert rowsig.signature(synth.expected_rows(cfg, m, max_pairs=1000)) == rowsig.signature(want)


def test_truth_matches_the_generated_reads():
    cfg = synth.SynthConfig(n_reads=50, read_len=400, genome_len=3000, ploidy=3, seed=9, len_sd=100.0, len_min=50, len_max=900)
    reads, tails = synth.generate_codes(cfg)
    t = synth.generate_truth(cfg)
    assert np.array_equal(tails, t.tails)
    for i, r in enumerate(reads):
        g = t.haps[t.hap_of[i]][t.starts[i]:t.starts[i] + t.lens[i]]
        assert np.array_equal(r, synth.revcomp_codes(g) if t.tails[i] else g)


def test_signature_sees_a_symmetric_loss():
    """What the property checks of round 2 could not see: a row dropped together with its strand mirror."""
    _, seqs, m, want = gu.ladder_case("ladder_cfg2_mini")
    lens = np.array([len(s) for s in seqs])
    sig = rowsig.signature(want)
    assert rowsig.signature(want[np.random.default_rng(0).permutation(len(want))]) == sig
    r = want[len(want) // 2]
    is_a = r[3] == lens[r[0]]
    mirror = (np.array([r[1] ^ 1, r[0] ^ 1, lens[r[1]] - r[5], lens[r[1]], 0, r[5]]) if is_a else
              np.array([r[0] ^ 1, r[1] ^ 1, lens[r[0]] - r[3], lens[r[0]] - r[2], 0, r[5]]))
    keep = ~((want == r).all(1) | (want == mirror).all(1))
    assert keep.sum() <= len(want) - 2
    assert rowsig.signature(want[keep]) != sig
    msg = rowsig.explain(want[keep], want)
    assert "missing" in msg and str(r.tolist()) in msg
    with pytest.raises(AssertionError):
        rowsig.assert_same_multiset(want[keep], want, "x")
