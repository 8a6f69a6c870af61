"""BASELINE.json configs 2, 3, 4 and 5 at FULL size on one GPU (config 2 also in test_gpu_e2e.py).

No CPU overlapper finishes these in seconds -- the reference would need 200-790 GB for its index
(SURVEY.md section 8a-2).  But the generator knows where every read comes from, so the COMPLETE expected multiset
is computable without any overlapper (``synth.expected_rows``: haplotype / start / strand of every read + prefix
sums over the haplotype difference masks; pinned to the reference's own output on all ladder goldens by
tests/test_synth_truth.py).  Every exact full-size call -- resident, sharded, streamed -- must return exactly that
multiset (contract: /root/reference/src/overlapper.cpp:64-116).  On top, the properties that hold at any size:

* every row is well formed and (sampled) a true match on the original read strings;
* A rows are unique per ordered pair;
* the multiset is closed under the strand mirror, compared as a checksum of row checksums;
* a closed NEIGHBOURHOOD of reads (a read, the reads it has rows with, and theirs, up to 400) gives exactly the
  rows the CPU oracle computes for those reads alone (rows depend only on the two reads involved);
* the union of three a-side shards is the same multiset as the whole-set call.

The same densities at 1 000 reads are reference goldens (cfg3_1k, cfg5_1k in test_gpu_parity.py::test_ladder_goldens).
"""
import os

import numpy as np
import pytest

import checker as ck
import rowsig
from rowsig import signature
from oracle import overlap_oracle as oo   # row helpers only
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("PHASM_SKIP_FULL") == "1", reason="full-size runs disabled")]

M = 1000


_EXPECTED = {}


def expected(cfg_name):
    """Signature of the generator-derived expected multiset (and the rows themselves for the smaller configs)."""
    if cfg_name not in _EXPECTED:
        rows = synth.expected_rows(synth.CONFIGS[cfg_name], M)
        _EXPECTED[cfg_name] = (signature(rows), rows if len(rows) < 10_000_000 else None)
    return _EXPECTED[cfg_name]


def assert_exact(rows, cfg_name, what):
    want_sig, want_rows = expected(cfg_name)
    got = signature(rows)
    if got != want_sig:
        if want_rows is None:
            want_rows = synth.expected_rows(synth.CONFIGS[cfg_name], M)
        raise AssertionError("%s %s: not the generator's expected multiset\n%s" % (cfg_name, what, rowsig.explain(rows, want_rows)))
    return got


def load(cfg_name, cfg=None):
    cfg = cfg or synth.CONFIGS[cfg_name]
    oriented = synth.oriented(synth.generate_reads(cfg))
    ov = ExactOverlapper(device=0)
    for n, s in oriented:
        ov.add_sequence(n, s)
    return cfg, oriented, ov


def neighbourhood(a, b, seed_read, cap=400):
    ring = {int(seed_read)}
    for _ in range(2):
        sel = np.isin(a, list(ring)) | np.isin(b, list(ring))
        more = np.unique(np.concatenate([a[sel], b[sel]]))
        for r in more.tolist():
            if len(ring) >= cap:
                break
            ring.add(r)
            ring.add(r ^ 1)   # both strands of a read: the mirror rows stay inside the set
    return np.array(sorted(ring), dtype=np.int64)


def check_properties(oriented, ov, arr, st, expect_rows_min):
    lens = np.array([len(s) for _, s in oriented], dtype=np.int64)
    rows = oo.struct_to_rows(arr)
    a, b, s, e, bs, be = (rows[:, k] for k in range(6))
    assert len(arr) == st["n_rows"] >= expect_rows_min
    assert (a != b).all() and (bs == 0).all() and (be >= M).all() and (e - s == be).all()
    is_a = e == lens[a]
    is_b = be == lens[b]
    assert (is_a | is_b).all()
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(arr), size=3000):
        assert oriented[a[i]][1][s[i]:e[i]] == oriented[b[i]][1][:be[i]]
    # A rows: one per ordered pair
    only_a = is_a & ~is_b
    key = (a[only_a] << 32) | b[only_a]
    assert len(np.unique(key)) == len(key)
    # strand-mirror closure (a row that is both A and B appears twice: once per family)
    both = rows[is_a & is_b]
    uniq, cnt = np.unique(both, axis=0, return_counts=True) if len(both) else (both, np.zeros(0, dtype=np.int64))
    assert (cnt % 2 == 0).all()
    half = uniq.repeat(cnt // 2, axis=0) if len(both) else both
    fa_ = np.concatenate([rows[only_a], half])
    fb_ = np.concatenate([rows[is_b & ~is_a], half])
    ma = np.stack([fa_[:, 1] ^ 1, fa_[:, 0] ^ 1, lens[fa_[:, 1]] - fa_[:, 5], lens[fa_[:, 1]], fa_[:, 4], fa_[:, 5]], axis=1)
    mb = np.stack([fb_[:, 0] ^ 1, fb_[:, 1] ^ 1, lens[fb_[:, 0]] - fb_[:, 3], lens[fb_[:, 0]] - fb_[:, 2], fb_[:, 4], fb_[:, 5]], axis=1)
    assert signature(np.concatenate([ma, mb])) == signature(rows)
    # a closed neighbourhood against the CPU oracle (in the checker process)
    seed_read = int(a[len(a) // 2])
    S = neighbourhood(a, b, seed_read)
    assert len(S) >= 40
    sub = [oriented[i][1] for i in S.tolist()]
    want = ck.oracle_overlaps(sub, M)
    inside = np.isin(a, S) & np.isin(b, S)
    got = rows[inside].copy()
    got[:, 0] = np.searchsorted(S, got[:, 0])
    got[:, 1] = np.searchsorted(S, got[:, 1])
    ck.assert_same_rows(oo.sort_rows(got), want, sub, M, "neighbourhood of read %d (%d reads)" % (seed_read, len(S)))
    assert len(want) > 100
    return signature(rows)


@pytest.mark.timeout(1500, method="thread")
@pytest.mark.parametrize("cfg_name,expect_rows", [("cfg3", 18_000_000), ("cfg5", 50_000_000)])
def test_full_size_large_configs(cfg_name, expect_rows):
    cfg, oriented, ov = load(cfg_name)
    res = ov.overlaps_result(M)
    arr = res.rows()
    st = ov.stats()
    res.free()
    assert st["wide_index"] == 1 and st["paired"] == 1 and st["bits_per_base"] == 2
    sig = check_properties(oriented, ov, arr, st, expect_rows)
    assert_exact(oo.struct_to_rows(arr), cfg_name, "resident call")     # the whole multiset, not a sample
    del arr
    # three a-side shards: the same multiset
    n, h1, h2 = 0, 0, 0
    for k in range(3):
        part = oo.struct_to_rows(ov.overlaps_shard_array(M, k, 3))
        pn, p1, p2 = signature(part)
        n, h1, h2 = n + pn, (h1 + p1) % (1 << 64), (h2 + p2) % (1 << 64)
        del part
    assert (n, h1, h2) == sig
    # the host-to-host call on a changed read set: streamed (wide index: the first TWO words of every read go ahead)
    ov.invalidate()
    res = ov.overlaps_to_host_result(M)
    st2 = ov.stats()
    got = signature(oo.struct_to_rows(res.rows_view()))
    res.free()
    assert st2["streamed"] == 1 and st2["wide_index"] == 1 and got == sig == expected(cfg_name)[0]
    # ... and once more on the same reads: pieces small enough for it run on their predicted candidate counts
    ov.invalidate()
    res = ov.overlaps_to_host_result(M)
    st3 = ov.stats()
    got3 = signature(oo.struct_to_rows(res.rows_view()))
    res.free()
    assert st3["streamed"] == 1 and got3 == sig, st3
    ov.close()


def test_full_size_cfg4_noise_emits_nothing_but_verifies_millions():
    """Config 4 = config 2 + 1 % substitutions: the reference's exact overlapper finds (next to) nothing on it --
    P[1000 error-free bases on both reads] ~ 2e-9 -- while half the anchors survive the filter and every one of
    several million candidates has to be rejected by the packed compare."""
    cfg, oriented, ov = load("cfg4")
    res = ov.overlaps_result(M)
    st = ov.stats()
    arr = res.rows()
    res.free()
    ov.close()
    assert st["n_candidates"] > 3_000_000 and st["paired"] == 1
    assert len(arr) == st["n_rows"]
    # the generator's expected multiset (noisy reads compared base by base): empty for this seed
    assert_exact(oo.struct_to_rows(arr), "cfg4", "exact path")
    assert len(arr) == expected("cfg4")[0][0] < 100


DP_E, DP_W = 400, 8     # what bench.py's cfg4 leg runs


def test_full_size_cfg4_banded_dp_equals_truth_and_the_cpu_dp(monkeypatch):
    """Config 4's defining feature at full size: the banded seed-extension DP (po_overlaps_ex(1000, 400, 8)) over
    100 k noisy 15 kb reads, 6.86 M candidates.  PARITY UNPINNED (the reference is exact, src/overlapper.cpp:28-150);
    what the rows are held against:

    * the CANDIDATES are the generator's truth: every (a, p, b) whose 32-base anchor is intact (synth.expected_candidates),
      count == po_stats.n_candidates, and every row sits on one of them;
    * substitution-only noise: the DP's cost on the main diagonal is at most the Hamming distance H of the two strings,
      so every candidate with H <= max_diff MUST have its A row -- H is computed for all 6.86 M candidates on the host;
    * the prediction "row iff H <= max_diff, ending on the main diagonal" is compared with the whole HIP output; wherever
      the two differ (two indels beating the substitutions between them, an end one base off the diagonal), for every
      candidate near the threshold, every candidate that can have a B row (p <= band) and a random 200 k sample, the CPU
      restatement's DP (oracle/extend_oracle.c:extend_one, on the listed pairs) decides -- HIP must equal it everywhere;
    * a closed 300-read neighbourhood equals oracle_overlaps_ex (which searches its anchors itself);
    * all three DP mappings agree on a 5 k-read slice."""
    from oracle import extend_oracle as eo          # pair-level helpers (pure functions on host arrays)
    cfg = synth.CONFIGS["cfg4"]
    codes, _ = synth.generate_codes(cfg)
    oriented = synth.oriented([("read%d" % i, synth.codes_to_ascii(r)) for i, r in enumerate(codes)])
    ov = ExactOverlapper(device=0)
    for n, sq in oriented:
        ov.add_sequence(n, sq)
    arr = ov.overlaps_ex_array(M, DP_E, DP_W)
    st = ov.stats()
    ov.close()
    rows = oo.struct_to_rows(arr)
    del arr
    assert st["dp_lanes"] == 2 and st["max_diff"] == DP_E and st["band"] == DP_W and st["n_rows"] == len(rows)

    c = synth.expected_candidates(cfg, M, 32, codes)
    a, b, p = c["a"], c["b"], c["p"]
    la, lb = c["lens"][a].astype(np.int64), c["lens"][b].astype(np.int64)
    rem = la - p
    assert len(a) == st["n_candidates"] > 6_000_000
    ax, ay = c["off"][a] + p, c["off"][b]
    H = eo.pair_hamming(c["cat"], ax, ay, np.minimum(rem, lb)).astype(np.int64)
    # every row sits on a candidate, every candidate has at most one A row (fixed-length reads: one anchor per ordered pair)
    key = lambda aa, bb, pp: (aa.astype(np.int64) << 40) | (bb.astype(np.int64) << 16) | pp.astype(np.int64)
    ckey = key(a, b, p)
    order = np.argsort(ckey)
    assert (np.diff(ckey[order]) > 0).all()
    pos = np.searchsorted(ckey[order], key(rows[:, 0], rows[:, 1], rows[:, 2]))
    assert (pos < len(ckey)).all() and (ckey[order][np.minimum(pos, len(ckey) - 1)] == key(rows[:, 0], rows[:, 1], rows[:, 2])).all(), "a row without a true anchor"
    cand_of_row = order[pos]
    is_a_row = rows[:, 3] == la[cand_of_row]
    # (a candidate with p <= band may also have a B row, and one that ends where a ends looks like an A row here;
    # which rows exactly is settled against the CPU DP below)
    assert (np.bincount(cand_of_row[is_a_row], minlength=len(a)) <= 1 + (p <= DP_W)).all()
    has_a = np.zeros(len(a), dtype=bool)
    has_a[cand_of_row[is_a_row]] = True
    assert has_a[H <= DP_E].all(), "a candidate within max_diff substitutions has no A row"
    assert st["dp_stopped"] <= int((~has_a).sum())      # a candidate that stopped early has no row

    # prediction from H alone, then the CPU DP wherever it matters
    want_ok_a = H <= DP_E
    want_ja = np.minimum(rem, lb)
    want_ok_b = np.zeros(len(a), dtype=bool)
    want_ib = np.zeros(len(a), dtype=np.int64)

    def settle(k):
        r = eo.extend_pairs(c["cat"], ax[k], rem[k], ay[k], lb[k], DP_E, DP_W).astype(np.int64)
        want_ok_a[k], want_ja[k], want_ok_b[k], want_ib[k] = r[:, 0] > 0, r[:, 1], r[:, 2] > 0, r[:, 3]

    def want_rows():
        ka, kb = np.nonzero(want_ok_a)[0], np.nonzero(want_ok_b)[0]
        z = np.zeros(len(a), dtype=np.int64)
        return np.concatenate([np.stack([a[ka], b[ka], p[ka], la[ka], z[ka], want_ja[ka]], 1),
                               np.stack([a[kb], b[kb], p[kb], p[kb] + want_ib[kb], z[kb], lb[kb]], 1)])

    rng = np.random.default_rng(4)
    special = np.unique(np.concatenate([np.nonzero((p <= DP_W) | (H > DP_E - 24))[0], rng.choice(len(a), 200_000, replace=False)]))
    settle(special)
    sampled_ok = want_ok_a[special]
    assert (sampled_ok == (H[special] <= DP_E))[H[special] <= DP_E].all()
    want = want_rows()
    if rowsig.signature(want) != rowsig.signature(rows):
        # the candidates on which prediction and HIP output differ: let the CPU DP speak there too, then it must be equal
        hw, hg = rowsig.row_hashes(want), rowsig.row_hashes(rows)
        wpos = np.searchsorted(ckey[order], key(want[:, 0], want[:, 1], want[:, 2]))
        differ = np.unique(np.concatenate([order[wpos][~np.isin(hw, hg)], cand_of_row[~np.isin(hg, hw)]]))
        assert 0 < len(differ) < len(a) // 500, "prediction and HIP differ on %d candidates" % len(differ)
        assert not np.isin(differ, special).any(), "HIP differs from the CPU DP on a candidate it has settled"
        settle(differ)
        want = want_rows()
    rowsig.assert_same_multiset(rows, want, "cfg4 banded DP at full size")
    assert int(want_ok_b.sum()) > 100 and int((want_ja != np.minimum(rem, lb))[want_ok_a].sum()) > 100    # both rare kinds occur

    # a closed neighbourhood against the restatement that finds its own anchors (checker process)
    S = neighbourhood(rows[:, 0], rows[:, 1], int(rows[len(rows) // 2, 0]), cap=300)
    sub = [oriented[i][1] for i in S.tolist()]
    want_n = ck.oracle_overlaps_ex(sub, M, DP_E, DP_W, anchor=32)
    inside = np.isin(rows[:, 0], S) & np.isin(rows[:, 1], S)
    got_n = rows[inside].copy()
    got_n[:, 0] = np.searchsorted(S, got_n[:, 0])
    got_n[:, 1] = np.searchsorted(S, got_n[:, 1])
    assert len(want_n) > 100 and np.array_equal(oo.sort_rows(got_n), want_n)
    del rows, want, c

    # both mappings of the DP on a 5 k-read slice of the same density
    cfg_s = synth.scaled(cfg, 5000)
    _, _, ov = load("cfg4", cfg_s)
    got = {}
    for kern in ("bits", "lanes", "wave"):
        monkeypatch.setenv("PHASM_DP_KERNEL", kern)
        got[kern] = oo.sort_rows(oo.struct_to_rows(ov.overlaps_ex_array(M, DP_E, DP_W)))
        assert ov.stats()["dp_lanes"] == {"bits": 2, "lanes": 1, "wave": 0}[kern]
    monkeypatch.delenv("PHASM_DP_KERNEL")
    ov.close()
    assert len(got["lanes"]) > 500_000 and np.array_equal(got["lanes"], got["wave"]) and np.array_equal(got["bits"], got["wave"])


@pytest.mark.parametrize("cfg_name", ["cfg2", "cfg4"])
def test_full_size_streamed_step_equals_the_resident_call(cfg_name, monkeypatch):
    """The bench's step at full size: po_overlaps_to_host on a changed read set takes the streamed form (reads
    uploaded in pieces under the kernels, reversed strand-mirror order).  Its rows are the multiset of the resident
    call -- whose rows pass the property checks above -- for the default cut and for a coarse and a fine one."""
    cfg, oriented, ov = load(cfg_name)
    res = ov.overlaps_result(M)
    arr = res.rows()
    st = ov.stats()
    res.free()
    if cfg_name == "cfg2":
        sig = check_properties(oriented, ov, arr, st, 6_000_000)
        assert_exact(oo.struct_to_rows(arr), "cfg2", "resident call")
        want = expected("cfg2")[1]
        assert np.array_equal(oo.sort_rows(oo.struct_to_rows(arr)), oo.sort_rows(want))    # and once as sorted arrays
    else:
        sig = signature(oo.struct_to_rows(arr))
    del arr
    for cuts in ("", "400,800", "50,100,150,200,300,400,500,600,700,800,900,950,980,995"):
        if cuts:
            monkeypatch.setenv("PHASM_STREAM_CUTS", cuts)
        ov.invalidate()
        res = ov.overlaps_to_host_result(M)
        st2 = ov.stats()
        got = signature(oo.struct_to_rows(res.rows_view()))
        res.free()
        assert st2["streamed"] == 1 and st2["paired"] == 1 and st2["n_rows"] == sig[0]
        # (candidate counts differ a little: a K-mer hit that does not verify has no counterpart on the mirror side)
        assert got == sig, "streamed step, cuts %r" % cuts
        # the same call again: every piece now runs on its PREDICTED candidate count (no host round trip) and ends in
        # the two-kernel tail -- the same multiset
        ov.invalidate()
        res = ov.overlaps_to_host_result(M)
        st3 = ov.stats()
        got3 = signature(oo.struct_to_rows(res.rows_view()))
        res.free()
        assert st3["streamed"] == 1 and got3 == sig, "streamed step with predicted counts, cuts %r" % cuts
        if cuts == "":
            # (the default schedule has 8 pieces on a handle's first streamed call and up to 12 from then on: the call
            # after the change of schedule has nothing to predict from, the one after that does)
            ov.invalidate()
            res = ov.overlaps_to_host_result(M)
            st4 = ov.stats()
            got4 = signature(oo.struct_to_rows(res.rows_view()))
            res.free()
            assert st4["streamed"] == 1 and got4 == sig
            if cfg_name == "cfg2":     # (pieces of fewer than 400 k candidates keep the host round trip)
                assert st4["n_predicted"] >= 6 and st4["fused_tail"] >= 6, st4
    ov.close()
