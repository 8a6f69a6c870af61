"""The checker side of the GPU parity tests, checked on the CPU: the checker process answers like the in-process
oracle, and the plain-Python contract evaluator that assert_same_rows uses to say which side of a mismatch is
wrong agrees with the reference's own outputs (every small golden)."""
from collections import Counter

import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo


@pytest.fixture(scope="module", autouse=True)
def _checker_process():
    mine = ck._sidecar is None   # (a session that runs GPU tests has started it already and keeps it)
    ck.start()
    yield
    if mine:
        ck.stop()


def test_checker_process_is_another_process_and_agrees_with_the_oracle():
    import os
    assert ck.sidecar().pid != os.getpid()
    for name, seqs, m, want in gu.all_small_cases()[:60]:
        assert np.array_equal(ck.oracle_overlaps(seqs, m), want), name
    rc, out, _ = ck.run(["/bin/echo", "hello"], capture_output=True, text=True)
    assert rc == 0 and out == "hello\n"


def test_contract_evaluator_matches_the_reference_goldens():
    """Every golden row has the multiplicity the evaluator computes; rows the reference does not emit get 0."""
    for name, seqs, m, want in gu.all_small_cases():
        seqs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        cnt = Counter(map(tuple, want.tolist()))
        for row, k in cnt.items():
            assert ck.expected_multiplicity(seqs, m, row) == k, (name, row)
        for row in list(cnt)[:5]:   # neighbours of true rows that are not rows
            a, b, s, e, bs, be = row
            if e - s > max(m, 1) and (a, b, s + 1, e, 0, be - 1) not in cnt:
                assert ck.expected_multiplicity(seqs, m, (a, b, s + 1, e, 0, be - 1)) == 0, (name, row)


def test_mismatch_is_a_failure_with_a_verdict():
    name, seqs, m, want = next(c for c in gu.all_small_cases() if len(c[3]) >= 2)
    short = want[1:]
    with pytest.raises(AssertionError) as ei:
        ck.assert_same_rows(want, short, seqs, m, "ctx")
    assert "CHECKER WRONG" in str(ei.value) and "HIP wrong on 0" in str(ei.value)
    with pytest.raises(AssertionError) as ei:
        ck.assert_same_rows(short, want, seqs, m, "ctx")
    assert "HIP WRONG" in str(ei.value) and "checker wrong on 0" in str(ei.value)
