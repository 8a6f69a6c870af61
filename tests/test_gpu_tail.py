"""The call's tail as ONE kernel (k_tail: rows per candidate, their prefix sum, the rows, the counters) and the single-pass
scan behind every prefix sum (k_ps_chain, decoupled look-back).

k_tail runs when the row buffer kept from an earlier call holds the worst case -- i.e. from the SECOND call on a handle
on -- so every golden is computed three times on one handle: classic tail first, fused tail after, same rows.  Reads
with hundreds of verified suffix-prefix hits (tandem repeats) make k_tail hand back to the classic kernels
(po_stats.tail_fallback); in a streamed step that abandons the step for the chunked form.  Rows are held against the
REFERENCE goldens (tests/golden/, /root/reference/src/overlapper.cpp:28-150)."""
import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo   # row helpers only
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu


def three_calls(seqs, m, call):
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    out = []
    for _ in range(3):
        rows = oo.sort_rows(oo.struct_to_rows(call(ov, m)))
        out.append((rows, ov.stats()))
    ov.close()
    return out


def whole(ov, m):
    return ov.overlaps_array(m)


def to_host(ov, m):
    ov.invalidate()
    res = ov.overlaps_to_host_result(m)
    arr = res.rows_view().copy()
    res.free()
    return arr


def sharded(ov, m):
    return np.concatenate([ov.overlaps_shard_array(m, k, 3) for k in range(3)])


@pytest.mark.parametrize("form", [whole, to_host, sharded], ids=["whole", "to_host", "sharded"])
def test_fused_tail_gives_the_golden_rows(form, monkeypatch):
    monkeypatch.setenv("PHASM_STREAM", "1")            # to_host: the streamed step wherever its preconditions hold
    monkeypatch.setenv("PHASM_STREAM_CUTS", "300,600,800")
    cases = gu.all_small_cases() + gu.repeats_cases() + [gu.ladder_case(n) for n in ("ladder_small", "ladder_varlen", "ladder_cfg2_mini")]
    n_fused = n_fallback = 0
    for name, seqs, m, want in cases:
        for call_no, (rows, st) in enumerate(three_calls(seqs, m, form)):
            ck.assert_same_rows(rows, want, seqs, m, "%s call %d (%s)" % (name, call_no, form.__name__))
            if call_no:
                n_fused += st["fused_tail"]
                n_fallback += st["tail_fallback"]
    assert n_fused > 20                                  # the fused kernel really ran ...
    if form is to_host:
        assert n_fallback > 0                            # ... and the tandem-repeat reads of repeats.npz sent it back


@pytest.mark.parametrize("home", ["1", "1 with 16-byte records", "0"])
def test_classic_and_fused_tails_emit_the_same_array(home, monkeypatch):
    """Same emission order, not just the same multiset: the prefix sums are exact either way -- and so is the host's
    expansion of the compact records (home = 1: po_overlaps_to_host brings one record per verified candidate home -- 8 bytes
    where the read set fits one word, 16 otherwise or with PHASM_HOME_PACK=0 -- and helper threads write the rows; home = 0:
    the device writes the rows and every one of them crosses PCIe).  The row array is the same bytes in all forms."""
    wide_records = home.endswith("16-byte records")
    home = home[0]
    if wide_records:
        monkeypatch.setenv("PHASM_HOME_PACK", "0")
    monkeypatch.setenv("PHASM_HOME", home)
    _, seqs, m, want = gu.ladder_case("cfg2_1k")
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    first = ov.overlaps_array(m)
    # (the record tail needs no kept worst-case row buffer: it runs from the first call on; the row tail from the second)
    assert ov.stats()["fused_tail"] == (1 if home == "1" else 0)
    assert ov.stats()["home_record_bytes"] == (0 if home == "0" else 16 if wide_records else 8)
    counters = {k: ov.stats()[k] for k in ("n_rows", "n_verified", "sum_overlap_bases", "verify_bytes_algo", "verify_bytes_exec")}
    assert counters["n_rows"] == len(want) and counters["sum_overlap_bases"] == int(want[:, 5].sum())   # (bend = l, bstart = 0)
    second = ov.overlaps_array(m)
    assert ov.stats()["fused_tail"] == 1 and ov.stats()["tail_fallback"] == 0
    monkeypatch.setenv("PHASM_TAIL_CLASSIC", "1")
    monkeypatch.setenv("PHASM_PS_CLASSIC", "1")
    third = ov.overlaps_array(m)
    assert ov.stats()["fused_tail"] == 0
    monkeypatch.delenv("PHASM_TAIL_CLASSIC")
    monkeypatch.delenv("PHASM_PS_CLASSIC")
    monkeypatch.setenv("PHASM_HOME", "0" if home == "1" else "1")     # ... and the other transfer form on the same handle
    fourth = ov.overlaps_array(m)
    # the byte counters of po_stats are the same whether the device summed them (rows) or the host did (records)
    assert {k: ov.stats()[k] for k in counters} == counters
    dev = ov.overlaps_result(m)                                       # the device-resident form (po_overlaps + po_result_rows)
    fifth = dev.rows()
    dev.free()
    ov.close()
    assert np.array_equal(first, second) and np.array_equal(first, third) and np.array_equal(first, fourth)
    assert np.array_equal(first, fifth)
    assert np.array_equal(oo.sort_rows(oo.struct_to_rows(first)), want)


def test_chained_scan_many_tiles(monkeypatch):
    """The look-back over more tiles than one window (64): a read set with > 64 * 4096 candidates, both scan forms."""
    _, seqs, m, want = gu.ladder_case("cfg2_1k")
    for classic in (False, True):
        if classic:
            monkeypatch.setenv("PHASM_PS_CLASSIC", "1")
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        for _ in range(3):
            monkeypatch.setenv("PHASM_NO_MIRROR", "1")      # every candidate verified and emitted itself: twice the items
            got = oo.sort_rows(oo.struct_to_rows(ov.overlaps_array(m)))
            assert np.array_equal(got, want)
        ov.close()


@pytest.mark.parametrize("scale", [None, "0.6"])
def test_streamed_pieces_with_a_predicted_candidate_count(scale, monkeypatch):
    """From the second streamed call on the same reads on, a piece does not wait for its candidate count: the kernels
    behind the counting pass are launched sized for the previous call's count and check the real one on the device.
    scale = 0.6: the prediction is too small -- every kernel must stand down, the step is abandoned for the chunked form,
    the rows are still the golden ones and the next call predicts again from a clean count."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "250,500,750")
    monkeypatch.setenv("PHASM_VERIFY_ORDER", "1")       # (the predicted path needs the locality order, which small inputs skip)
    for name in ("ladder_cfg2_mini", "cfg2_1k", "cfg3_1k", "ladder_varlen"):
        _, seqs, m, want = gu.ladder_case(name)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        for call_no in range(4):
            if scale and call_no == 2:
                monkeypatch.setenv("PHASM_PRED_SCALE", scale)
            else:
                monkeypatch.delenv("PHASM_PRED_SCALE", raising=False)
            rows = oo.sort_rows(oo.struct_to_rows(to_host(ov, m)))
            st = ov.stats()
            ck.assert_same_rows(rows, want, seqs, m, "%s call %d" % (name, call_no))
            if scale and call_no == 2:
                assert st["streamed"] == 0 and st["tail_fallback"] >= 1, (name, st)
            else:
                assert st["streamed"] == 1, (name, call_no, st)
                # (call 0 has nothing to predict from; the call after an abandoned step neither)
                assert (st["n_predicted"] > 0) == (call_no in ((1, 2, 3) if not scale else (1,))), (name, call_no, st)
            assert st["n_candidates"] > 0 and st["n_rows"] == len(want)
        ov.close()


def test_classic_tail_switch_on_a_call_that_would_predict(monkeypatch):
    """PHASM_TAIL_CLASSIC=1 on the SECOND streamed call of a handle: the first call left candidate counts to predict from,
    but a predicted piece has nothing but the fused tail -- with the switch set no piece may predict, and the classic
    kernels must see the real count (they once ran unguarded over the predicted capacity)."""
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_STREAM_CUTS", "250,500,750")
    monkeypatch.setenv("PHASM_VERIFY_ORDER", "1")
    for name in ("cfg2_1k", "ladder_varlen"):
        _, seqs, m, want = gu.ladder_case(name)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        for call_no in range(4):
            if call_no in (1, 2):
                monkeypatch.setenv("PHASM_TAIL_CLASSIC", "1")
            else:
                monkeypatch.delenv("PHASM_TAIL_CLASSIC", raising=False)
            rows = oo.sort_rows(oo.struct_to_rows(to_host(ov, m)))
            st = ov.stats()
            ck.assert_same_rows(rows, want, seqs, m, "%s call %d" % (name, call_no))
            assert st["streamed"] == 1 and st["n_rows"] == len(want)
            if call_no in (1, 2):
                assert st["n_predicted"] == 0 and st["fused_tail"] == 0, (name, call_no, st)
            elif call_no == 3:
                assert st["n_predicted"] > 0 and st["fused_tail"] > 0, (name, call_no, st)
        ov.close()


def test_predicted_pieces_fuzz(monkeypatch):
    """Several streamed calls per handle with the cuts, min_length and the read set changing in between: a prediction
    must be used only for the same reads, cuts and min_length, and a wrong one must never show in the rows."""
    import os
    monkeypatch.setenv("PHASM_STREAM", "1")
    monkeypatch.setenv("PHASM_VERIFY_ORDER", "1")
    rng = np.random.default_rng(20260)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n_pred = 0
    for trial in range(int(os.environ.get("PHASM_SOAK_TRIALS", "8"))):
        glen = int(rng.integers(800, 5000))
        genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        seqs = []
        for _ in range(int(rng.integers(8, 70))):
            ln = int(rng.integers(40, max(60, glen // 3)))
            s = int(rng.integers(0, glen - ln))
            r = genome[s:s + ln]
            seqs += [r, r.translate(comp)[::-1]]
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        m = int(rng.choice([33, 40, 70]))
        cuts = "300,600"
        for call in range(6):
            what = int(rng.integers(0, 5))
            if what == 0:
                cuts = ",".join(str(c) for c in sorted(set(int(c) for c in rng.integers(50, 950, size=int(rng.integers(1, 6))))))
            elif what == 1:
                m = int(rng.choice([33, 40, 70]))
            elif what == 2:       # the read set grows: the old counts are too small for the pieces they belonged to
                for _ in range(int(rng.integers(1, 12))):
                    ln = int(rng.integers(40, max(60, glen // 3)))
                    s = int(rng.integers(0, glen - ln))
                    r = genome[s:s + ln]
                    for x in (r, r.translate(comp)[::-1]):
                        ov.add_sequence("r%d" % len(seqs), x)
                        seqs.append(x)
            monkeypatch.setenv("PHASM_STREAM_CUTS", cuts)
            if what == 3:
                monkeypatch.setenv("PHASM_PRED_SCALE", "0.5")
            else:
                monkeypatch.delenv("PHASM_PRED_SCALE", raising=False)
            got = oo.sort_rows(oo.struct_to_rows(to_host(ov, m)))
            st = ov.stats()
            n_pred += st["n_predicted"]
            ck.assert_same_rows(got, ck.oracle_overlaps(seqs, m), seqs, m, "trial %d call %d (%d, cuts %s, m %d)" % (trial, call, what, cuts, m))
        ov.close()
    assert n_pred > 10
