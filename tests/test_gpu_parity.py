"""Parity tests proper: the HIP path, called through the C ABI (ctypes shim), against the
golden vectors produced by the compiled reference and against the CPU oracle on fresh seeded
inputs.  Bit-exact: integer rows compared as sorted multisets (the reference's row order is
implementation-defined, overlapper.cpp:30,:68)."""
import os

import numpy as np
import pytest

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo   # row helpers only (sort_rows / struct_to_rows): the oracle itself runs in the
from phasm_amd import synth               # checker process (tests/checker.py), never in this one
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu

_last = {}


def hip_rows(seqs, m, shard=None):
    _last.update(seqs=seqs, m=m)
    guard = ck.GuardedReads(seqs)   # (the reads in a read-only mapping; pointers into it go to po_add_sequence)
    ov = ExactOverlapper()
    guard.add_all(ov)
    if shard is None:
        arr = ov.overlaps_array(m)
    else:
        arr = np.concatenate([ov.overlaps_shard_array(m, k, shard) for k in range(shard)])
    st = ov.stats()
    ov.close()
    guard.verify_and_close()
    return oo.sort_rows(oo.struct_to_rows(arr)), st


def same(got, want, ctx=""):
    """HIP rows == expected rows as sorted multisets; a mismatch is a failure that says which side is wrong
    (tests/checker.py: the contract evaluated on the reads of the last hip_rows call)."""
    ck.assert_same_rows(got, want, _last.get("seqs"), _last.get("m", 1), ctx)


def test_toy_and_adversarial_goldens():
    cases = gu.all_small_cases()
    bits_seen = set()
    for name, seqs, m, want in cases:
        got, st = hip_rows(seqs, m)
        same(got, want, name)
        bits_seen.add(st["bits_per_base"])
    assert 2 in bits_seen  # (short reads with N / lower case stay 2-bit + exception records)


def test_repeats_goldens():
    for name, seqs, m, want in gu.repeats_cases():
        got, _ = hip_rows(seqs, m)
        same(got, want, name)


@pytest.mark.parametrize("name", gu.LADDER_NAMES)
def test_ladder_goldens(name):
    _, seqs, m, want = gu.ladder_case(name)
    got, st = hip_rows(seqs, m)
    same(got, want)
    assert st["n_rows"] == len(want)
    assert st["paired"] == 1  # both strands were added: the strand-mirror shortcut is active


@pytest.mark.parametrize("name", ["ladder_varlen", "ladder_cfg2_mini"])
def test_mirror_shortcut_off_gives_the_same_rows(name, monkeypatch):
    _, seqs, m, want = gu.ladder_case(name)
    monkeypatch.setenv("PHASM_NO_MIRROR", "1")
    got, st = hip_rows(seqs, m)
    assert st["paired"] == 0
    same(got, want)


def test_pairing_is_detected_not_assumed():
    """An even number of reads that are NOT strand pairs (one base off) must not be mirrored."""
    _, seqs, m, _ = gu.ladder_case("ladder_small")
    seqs = list(seqs)
    s = bytearray(seqs[41])
    s[7] = ord("A") if s[7] != ord("A") else ord("C")
    seqs[41] = bytes(s)
    got, st = hip_rows(seqs, m)
    assert st["paired"] == 0
    same(got, ck.oracle_overlaps(seqs, m))
    # odd lengths / odd tail words still pair up
    rng = np.random.default_rng(8)
    reads = []
    for ln in (1, 2, 31, 32, 33, 63, 64, 65, 100, 257):
        r = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=ln))
        reads += [r, r.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]]
    got, st = hip_rows(reads, 1)
    assert st["paired"] == 1
    same(got, ck.oracle_overlaps(reads, 1))


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_shard_union_equals_whole(nshards):
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    got, _ = hip_rows(seqs, m, shard=nshards)
    same(got, want)
    for name, seqs, m, want in gu.repeats_cases()[:1]:
        got, _ = hip_rows(seqs, m, shard=nshards)
        same(got, want, name)


def test_byte_mode_matches_oracle_on_mixed_alphabet():
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    genome = alpha[rng.integers(0, len(alpha), size=6000)].tobytes()
    seqs = []
    for _ in range(150):
        ln = int(rng.integers(30, 900))
        st = int(rng.integers(0, len(genome) - ln))
        seqs.append(genome[st:st + ln])
    for m in (8, 9, 40):
        got, st = hip_rows(seqs, m)
        assert st["bits_per_base"] == 8
        same(got, ck.oracle_overlaps(seqs, m), m)


@pytest.mark.parametrize("stream", ["0", "1"])
def test_repeated_calls_and_incremental_adds(stream, monkeypatch):
    """Row ORDER is not part of the contract (the reference's is std::unordered_map iteration order,
    /root/reference/src/overlapper.cpp:30,:68).  What this library promises: the same call FORM on the same reads gives
    the same array -- the resident form (a-major in insertion order) and the streamed form (pieces in index order, a
    pair's rows attached to its later read) each have their own order, and a changed read set makes overlaps_array's
    host-to-host call take the streamed form once (stream = 1 forces it for this small input)."""
    monkeypatch.setenv("PHASM_STREAM", stream)
    _, seqs, m, want = gu.ladder_case("ladder_small")
    ov = ExactOverlapper()
    half = len(seqs) // 2
    for i, s in enumerate(seqs[:half]):
        ov.add_sequence("r%d" % i, s)
    first = oo.sort_rows(oo.struct_to_rows(ov.overlaps_array(m)))
    same(first, ck.oracle_overlaps(seqs[:half], m))
    for i, s in enumerate(seqs[half:]):
        ov.add_sequence("r%d" % (half + i), s)
    a = ov.overlaps_array(m)     # (the read set changed: with PHASM_STREAM=1 this call is the streamed form ...)
    st_a = ov.stats()["streamed"]
    b = ov.overlaps_array(m)     # (... and this one the resident, chunked form)
    c2 = ov.overlaps_array(m)
    assert st_a == int(stream) and ov.stats()["streamed"] == 0
    assert np.array_equal(b, c2)  # the same form twice: the same rows in the same order
    if stream == "0":
        assert np.array_equal(a, b)
    for x in (a, b):
        assert np.array_equal(oo.sort_rows(oo.struct_to_rows(x)), want)
    # two handles, the same reads, the same (streamed or not) first call: the same array
    ov2 = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov2.add_sequence("r%d" % i, s)
    a2 = ov2.overlaps_array(m)
    ov2.close()
    assert sorted(map(tuple, a2.tolist())) == sorted(map(tuple, a.tolist()))
    if stream == "0":
        assert np.array_equal(a2, a)
    # a different min_length on the same handle (index is rebuilt per call, overlapper.cpp:33-36)
    c = oo.sort_rows(oo.struct_to_rows(ov.overlaps_array(m * 3)))
    same(c, ck.oracle_overlaps(seqs, m * 3))
    ov.close()


def test_tuple_api_matches_reference_shape():
    ov = ExactOverlapper()
    ov.add_sequence("r1+", "AAACCCGGGTTT")
    ov.add_sequence("r2+", "GGGTTTACGTAC")
    ov.add_sequence("r3+", "CCCGGG")
    rows = ov.overlaps(3)
    assert isinstance(rows, list) and all(isinstance(r, tuple) for r in rows)
    assert sorted(rows) == sorted([("r1+", "r3+", 3, 9, 0, 6), ("r1+", "r2+", 6, 12, 0, 6),
                                   ("r3+", "r2+", 3, 6, 0, 3)])
    with pytest.raises(TypeError):
        ov.overlaps(-1)
    assert ExactOverlapper().overlaps(5) == []  # no reads: [] (reference: undefined behaviour)


def test_midsize_against_oracle():
    """cfg2 density at 2 000 reads (4 000 oriented): same coverage per haplotype as cfg2."""
    cfg = synth.scaled(synth.CONFIGS["cfg2"], 2000)
    seqs = [s for _, s in synth.oriented(synth.generate_reads(cfg))]
    got, st = hip_rows(seqs, 1000)
    want = ck.oracle_overlaps(seqs, 1000)
    assert len(want) > 100_000
    same(got, want)
    assert st["bits_per_base"] == 2 and st["kmer"] == 32


def test_long_chain_many_identical_reads():
    rng = np.random.default_rng(3)
    base = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=300))
    seqs = [base] * 40 + [base[50:] + b"ACGTACGT", b"TTTT" + base[:200]]
    got, _ = hip_rows(seqs, 100)
    same(got, ck.oracle_overlaps(seqs, 100))


@pytest.mark.parametrize("nshards", [1, 2, 5])
def test_compact_candidates_then_expand_equals_rows(nshards):
    """The multi-GPU exchange form: per-shard verified candidates (16 B), concatenated in shard
    order, expanded by po_expand -> exactly the po_overlaps rows as a multiset, and the very same row array
    for every shard count (sharded calls pick the canonical member of a strand-mirror pair by a scrambled
    read order, whole-set calls by index order: same rows, different emission order)."""
    import torch
    from phasm_amd.dist import _result_to_tensor
    for case in ("ladder_varlen", "ladder_cfg2_mini"):
        _, seqs, m, want = gu.ladder_case(case)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        whole = ov.overlaps_array(m)
        dev = torch.device("cuda", 0)

        def expanded(ns):
            parts = []
            for k in range(ns):
                res = ov.candidates_result(m, k, ns)
                parts.append(_result_to_tensor(res, 4, dev))
                res.free()
            merged = torch.cat(parts, dim=0).contiguous()
            res = ov.expand_result(merged.data_ptr(), merged.shape[0])
            rows = res.rows()
            res.free()
            return merged, rows

        merged, got = expanded(nshards)
        assert merged.shape[0] < len(whole)          # paired mode: one candidate per mirror pair
        assert np.array_equal(oo.sort_rows(oo.struct_to_rows(got)), oo.sort_rows(oo.struct_to_rows(whole)))
        assert np.array_equal(oo.sort_rows(oo.struct_to_rows(got)), want)
        if nshards > 1:
            _, other = expanded(5)
            same(got, other)  # row for row, whatever the shard count
        # garbage in -> loud failure, not garbage rows
        bad = merged.clone()
        bad[0, 0] = 10 ** 9
        with pytest.raises(ValueError):
            ov.expand_result(bad.data_ptr(), bad.shape[0])
        ov.close()


@pytest.mark.parametrize("waves", ["1", "3", "16"])
def test_scan_pipeline_many_tiles_per_wave(waves, monkeypatch):
    """The scan kernel is a hand-scheduled load pipeline (counted s_waitcnt, in-flight landing
    registers).  Few waves per workgroup = many tiles per wave = the steady state of that pipeline;
    a mis-counted wait shows up as run-to-run differences, so run it repeatedly."""
    monkeypatch.setenv("PHASM_SCAN_WAVES", waves)
    _, seqs, m, want = gu.ladder_case("ladder_cfg2_mini")
    for _ in range(3):
        got, _ = hip_rows(seqs, m)
        same(got, want)
    monkeypatch.setenv("PHASM_NO_MIRROR", "1")
    got, _ = hip_rows(seqs, m)
    same(got, want)


@pytest.mark.parametrize("mult,waves", [("1.5", "16"), ("1.5", "2"), ("40", "16")])
def test_crowded_and_sparse_anchor_table(mult, waves, monkeypatch, capfd):
    """The narrow anchor table is probed in aligned groups of four slots; a probe whose group is taken by other
    keys goes to the leftover list (k_scan_fixup) or, when that is full, is resolved in place.  1.5 slots per key
    crowds the groups (2000 keys in 4096 slots: about one group in ten is full), 40 leaves them almost empty."""
    monkeypatch.setenv("PHASM_TABLE_MULT", mult)
    monkeypatch.setenv("PHASM_SCAN_WAVES", waves)
    monkeypatch.setenv("PHASM_DEBUG_LEFT", "1")
    for name in ("cfg2_1k", "cfg1_full"):
        _, seqs, m, want = gu.ladder_case(name)
        got, _ = hip_rows(seqs, m)
        same(got, want)
    err = capfd.readouterr().err
    deferred = [int(ln.split("deferred")[1].split()[0]) for ln in err.splitlines() if ln.startswith("[left]")]
    assert deferred, "PHASM_DEBUG_LEFT printed nothing"
    if mult == "1.5":
        assert max(deferred) > 0      # the crowded table did exercise the leftover path


def test_full_leftover_list_resolves_in_place(tmp_path):
    """A scan wave whose leftover list is full settles hard positions on the spot (synchronous probes inside the
    pipeline).  The list holds 2048 entries per wave, which no test input fills: build the library with room for 8
    (hipcc is on the GPU box) and run two goldens through it in a child process."""
    import shutil
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libphasm_overlap_cap8.so")
    # (both children are started by the checker process: this one has the GPU open and does not fork)
    rc, _, err = ck.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DPO_LEFT_CAP=8", "-o", lib,
                         os.path.join(root, "phasm_amd", "csrc", "c_api.hip")], timeout=900, capture_output=True, text=True)
    assert rc == 0, err[-3000:]
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import golden_utils as gu\n"
        "from test_gpu_parity import hip_rows\n"
        "for name in ('cfg2_1k', 'ladder_cfg2_mini', 'cfg1_full'):\n"
        "    _, seqs, m, want = gu.ladder_case(name)\n"
        "    got, _ = hip_rows(seqs, m)\n"
        "    assert np.array_equal(got, want), name\n"
        "print('CAP8 OK')\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ, PHASM_LIB=lib, PHASM_TABLE_MULT="1.5", PHASM_SCAN_WAVES="2", PHASM_DEBUG_LEFT="1")
    rc, stdout, stderr = ck.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert rc == 0 and "CAP8 OK" in stdout, stdout[-2000:] + stderr[-3000:]
    caps = [ln for ln in stderr.splitlines() if ln.startswith("[left]")]
    assert caps and all("cap 8" in ln for ln in caps), caps[:3]       # the variant library was the one that ran
    assert any("max 8 per wave" in ln for ln in caps), caps[:6]       # and some wave's list did fill up


def test_long_reads_take_the_global_verify_path():
    """Reads longer than the 64 KB LDS staging limit (262 144 bases at 2 bit) are compared straight
    from global memory."""
    rng = np.random.default_rng(21)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=700_000)].tobytes()
    reads = [genome[0:400_000], genome[150_000:600_000], genome[390_000:700_000], genome[100_000:130_000],
             genome[399_000:400_500]]
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for r in reads:
        seqs += [r, r.translate(rc)[::-1]]
    got, st = hip_rows(seqs, 1000)
    assert st["paired"] == 1
    want = ck.oracle_overlaps(seqs, 1000)
    assert len(want) >= 8
    same(got, want)


def test_wide_index_matches_goldens(monkeypatch):
    """Large-read-set flavour of the scan (W K-mers per read, word-aligned probes), forced on small
    goldens: ladders (2-bit), low-complexity repeats, shards, mirror off, and 8-bit reads."""
    monkeypatch.setenv("PHASM_INDEX", "wide")
    for name in gu.LADDER_NAMES:
        _, seqs, m, want = gu.ladder_case(name)
        got, st = hip_rows(seqs, m)
        assert st["wide_index"] == 1, name
        same(got, want, name)
    for name, seqs, m, want in gu.repeats_cases():
        got, st = hip_rows(seqs, m)
        assert st["wide_index"] == (1 if m >= 63 else 0)   # needs min_length >= 2W-1
        same(got, want, name)
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    got, _ = hip_rows(seqs, m, shard=3)
    same(got, want)
    monkeypatch.setenv("PHASM_NO_MIRROR", "1")
    got, st = hip_rows(seqs, m)
    assert st["paired"] == 0
    same(got, want)
    # 8-bit reads: W = 8, needs min_length >= 15
    rng = np.random.default_rng(6)
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    genome = alpha[rng.integers(0, len(alpha), size=5000)].tobytes()
    seqs8 = [genome[st:st + ln] for st, ln in zip(rng.integers(0, 4000, size=120), rng.integers(30, 900, size=120))]
    for m8 in (15, 40):
        got, st = hip_rows(seqs8, m8)
        assert st["bits_per_base"] == 8 and st["wide_index"] == 1
        same(got, ck.oracle_overlaps(seqs8, m8), m8)


@pytest.mark.parametrize("window", ["1", "4", "16"])
def test_wide_index_window_minimisers(window, monkeypatch):
    """The wide index probes only the window minimisers among a's words (kernels.hip.h WideEnc; window 16 from
    min_length 543 on, 4 from 159 on): every window must give the reference's rows -- ladders (variable lengths,
    containments, three and four haplotypes), tandem repeats (hundreds of verified hits per pair, ties inside windows),
    shards, mirror off, and read sets whose min_length sits exactly on a window's threshold."""
    monkeypatch.setenv("PHASM_INDEX", "wide")
    monkeypatch.setenv("PHASM_WIDE_WINDOW", window)
    for name in gu.LADDER_NAMES:
        _, seqs, m, want = gu.ladder_case(name)
        got, st = hip_rows(seqs, m)
        assert st["wide_index"] == 1, name
        same(got, want, name)
    for name, seqs, m, want in gu.repeats_cases():
        got, st = hip_rows(seqs, m)
        same(got, want, name)
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    got, _ = hip_rows(seqs, m, shard=3)
    same(got, want)
    monkeypatch.setenv("PHASM_NO_MIRROR", "1")
    got, st = hip_rows(seqs, m)
    same(got, want)
    monkeypatch.delenv("PHASM_NO_MIRROR")
    # min_length on and around the thresholds (W ww + W - 1 = 159 and 543), reads barely longer than that, low complexity
    rng = np.random.default_rng(41)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    for trial in range(12):
        glen = int(rng.integers(700, 5000))
        if trial % 3 == 2:
            unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(2, 50))))
            g = bytearray((unit * (glen // len(unit) + 1))[:glen])
            for pos in rng.integers(0, glen, size=glen // 80):
                g[pos] = b"ACGT"[rng.integers(4)]
            genome = bytes(g)
        else:
            genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        m = int(rng.choice([158, 159, 160, 542, 543, 544, 600]))
        seqs = []
        for _ in range(int(rng.integers(6, 50))):
            ln = min(glen, int(rng.choice([m, m + 1, m + 31, m + 32, 2 * m, int(rng.integers(m, 4 * m))])))
            st0 = int(rng.integers(0, glen - ln + 1))
            r = genome[st0:st0 + ln]
            seqs += [r, r.translate(rc)[::-1]]
        got, st = hip_rows(seqs, m)
        same(got, ck.oracle_overlaps(seqs, m), "trial %d m %d window %s" % (trial, m, window))


def test_wide_index_midsize_against_oracle(monkeypatch):
    monkeypatch.setenv("PHASM_INDEX", "wide")
    cfg = synth.scaled(synth.CONFIGS["cfg2"], 2000)
    seqs = [s for _, s in synth.oriented(synth.generate_reads(cfg))]
    got, st = hip_rows(seqs, 1000)
    assert st["wide_index"] == 1
    same(got, ck.oracle_overlaps(seqs, 1000))


@pytest.mark.parametrize("index", ["narrow", "wide"])
def test_seeded_fuzz_against_oracle(index, monkeypatch):
    """Random read sets the goldens do not cover: mixed lengths around the word/tile boundaries
    (31..33, 63..65, 2047..2049 bases), tandem repeats, duplicated and contained reads, both strands or
    not, min_length from 1 up, both index flavours -- HIP rows must equal the CPU oracle's."""
    monkeypatch.setenv("PHASM_INDEX", index)
    rng = np.random.default_rng(int(os.environ.get("PHASM_FUZZ_SEED", "2024")))
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    for trial in range(int(os.environ.get("PHASM_FUZZ_TRIALS", "40"))):
        glen = int(rng.integers(300, 6000))
        if rng.random() < 0.3:   # low-complexity genome
            unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(1, 12))))
            genome = (unit * (glen // len(unit) + 1))[:glen]
            genome = bytearray(genome)
            for pos in rng.integers(0, glen, size=glen // 50):
                genome[pos] = b"ACGT"[rng.integers(4)]
            genome = bytes(genome)
        else:
            genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen))
        special = [31, 32, 33, 63, 64, 65, 127, 128, 129, 2047, 2048, 2049]
        reads = []
        for _ in range(int(rng.integers(4, 60))):
            ln = int(rng.choice(special)) if rng.random() < 0.3 else int(rng.integers(1, 2500))
            ln = min(ln, glen)
            st = int(rng.integers(0, glen - ln + 1))
            r = genome[st:st + ln]
            if rng.random() < 0.5:
                r = r.translate(rc)[::-1]
            reads.append(r)
            if rng.random() < 0.1:
                reads.append(r)                       # exact duplicate
        if rng.random() < 0.6:                        # both strands, as the CLI adds them
            seqs = []
            for r in reads:
                seqs += [r, r.translate(rc)[::-1]]
        else:
            seqs = reads
        m = int(rng.choice([1, 2, 5, 31, 32, 33, 62, 63, 64, 100, 500]))
        got, st = hip_rows(seqs, m)
        want = ck.oracle_overlaps(seqs, m)
        same(got, want, "trial %d m %d reads %d wide %d paired %d" % (trial, m, len(seqs), st["wide_index"], st["paired"]))
        # the sharded form of the same call (scrambled canonical order, every foreign read a repeat suspect)
        got3, _ = hip_rows(seqs, m, shard=3)
        same(got3, want, (trial, m, len(seqs), "3 shards"))


def test_sparse_non_acgt_bytes_stay_on_the_2bit_path():
    """A few N / IUPAC / lower-case bytes must not push the whole read set onto the slow 8-bit path:
    they are kept as exception records beside the 2-bit codes and compared after the packed compare.
    Byte equality like the reference: N matches N only, 'a' is not 'A'."""
    rng = np.random.default_rng(77)
    cfg = synth.SynthConfig(n_reads=150, read_len=3000, genome_len=25_000, ploidy=2, snp=0.004, seed=31)
    reads = [bytearray(s) for _, s in synth.generate_reads(cfg)]
    genome_like = []
    # plant the SAME exceptional bytes at the same genome positions in overlapping reads by editing a
    # shared genome instead: regenerate reads from a genome that already contains N / R / lower case
    g = bytearray(bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=25_000)))
    for pos in rng.integers(0, len(g), size=60):
        g[pos] = rng.choice(list(b"NNNRYacgtn"))
    g = bytes(g)
    plain = []
    for _ in range(160):
        st = int(rng.integers(0, len(g) - 3000))
        plain.append(g[st:st + 3000])
    # a read that differs from another only in an exceptional byte (N vs A): must NOT match there
    twin = bytearray(plain[0])
    k = next((i for i, c in enumerate(twin) if c not in b"ACGT"), None)
    if k is not None:
        twin[k] = ord("A")
        plain.append(bytes(twin))
    from phasm_amd.io.fasta import reverse_complement
    seqs = []
    for r in plain:
        seqs += [r, reverse_complement(r)]
    for m in (500, 1000):
        got, st = hip_rows(seqs, m)
        assert st["bits_per_base"] == 2          # exceptions, not the 8-bit path
        assert st["paired"] == 1                 # strand pairs recognised although they contain N / R / Y
        want = ck.oracle_overlaps(seqs, m)
        assert len(want) > 500
        same(got, want, m)
    # unpaired use + mirror off give the same rows
    got, st = hip_rows(seqs[:-1], 500)
    assert st["paired"] == 0
    same(got, ck.oracle_overlaps(seqs[:-1], 500))
    # a pair that is NOT a reverse complement at an exceptional byte only
    bad = list(seqs)
    j = next(i for i, s_ in enumerate(bad) if i % 2 == 1 and any(c not in b"ACGT" for c in s_))
    b = bytearray(bad[j])
    kk = next(i for i, c in enumerate(b) if c not in b"ACGT")
    b[kk] = ord("N") if b[kk] != ord("N") else ord("R")
    bad[j] = bytes(b)
    got, st = hip_rows(bad, 500)
    assert st["paired"] == 0
    same(got, ck.oracle_overlaps(bad, 500))


def test_tandem_repeats_many_hits_per_pair():
    """Reads from a short-period tandem repeat: every read's prefix recurs inside itself and inside the
    others hundreds of times, so each ordered pair has hundreds of verified A candidates of which only
    the longest may be reported (plus every containment occurrence).  The duplicate resolution is a
    hash table keyed by (a, b), not a quadratic look-back."""
    rng = np.random.default_rng(5)
    unit = b"ACGGTCA"
    genome = bytearray(unit * 600)
    for pos in rng.integers(0, len(genome), size=12):      # a few point differences
        genome[pos] = b"ACGT"[rng.integers(4)]
    genome = bytes(genome)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for _ in range(50):
        ln = int(rng.integers(700, 1600))
        st = int(rng.integers(0, len(genome) - ln))
        r = genome[st:st + ln]
        seqs += [r, r.translate(rc)[::-1]]
    for m in (64, 300):
        got, st = hip_rows(seqs, m)
        want = ck.oracle_overlaps(seqs, m)
        assert st["n_candidates"] > 20 * len(want) > 0      # far more hits than rows
        same(got, want, m)


def test_expand_skips_all_zero_padding():
    """The candidate exchange travels in equal-sized slots (no all-gatherv in RCCL): po_expand ignores
    all-zero entries anywhere in the array and still rejects every other impossible entry."""
    import torch
    from phasm_amd.dist import _result_to_tensor
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    dev = torch.device("cuda", 0)
    parts = []
    for k in range(3):
        res = ov.candidates_result(m, k, 3)
        t = _result_to_tensor(res, 4, dev)
        res.free()
        parts += [t, torch.zeros((17 * (k + 1), 4), dtype=torch.int32, device=dev)]
    merged = torch.cat(parts).contiguous()
    res = ov.expand_result(merged.data_ptr(), merged.shape[0])
    got = oo.sort_rows(oo.struct_to_rows(res.rows()))
    res.free()
    same(got, want)
    bad = merged.clone()
    bad[-1, 3] = 1                      # a = b = 0 with a type: not padding, not possible
    with pytest.raises(ValueError):
        ov.expand_result(bad.data_ptr(), bad.shape[0])
    ov.close()


@pytest.mark.parametrize("name", ["ladder_small", "ladder_varlen", "ladder_cfg2_mini", "cfg2_1k"])
def test_verify_locality_order_forced_on_small_sets(name, monkeypatch):
    """The locality order of the verify grid (label, LDS counting sort, XCD-aware mapping) normally switches on
    only for big calls; forced on here so that the goldens also cover it -- with both index flavours and with
    the mirror shortcut off (labels are then plain indices)."""
    _, seqs, m, want = gu.ladder_case(name)
    monkeypatch.setenv("PHASM_VERIFY_ORDER", "1")
    got, _ = hip_rows(seqs, m)
    same(got, want)
    monkeypatch.setenv("PHASM_INDEX", "wide")
    got, st = hip_rows(seqs, m)
    assert st["wide_index"] == 1
    same(got, want)
    monkeypatch.setenv("PHASM_NO_MIRROR", "1")
    got, st = hip_rows(seqs, m)
    assert st["paired"] == 0
    same(got, want)


@pytest.mark.skipif(not oo.have_reference(), reason="oracle/_ref/ref_overlapper not built (make -C oracle ref)")
def test_live_against_the_reference_binary_on_fresh_data():
    """Beyond the committed goldens: the compiled reference itself (oracle/_ref, it travels with the repository
    snapshot) on read sets it has not been run on before -- config-2 density and a variable-length diploid set
    with containments -- against the HIP rows, whole-set and as three shards."""
    cases = [(synth.SynthConfig(**{**synth.scaled(synth.CONFIGS["cfg2"], 700).__dict__, "seed": 4242}), 1000),
             (synth.SynthConfig(n_reads=500, read_len=6000, genome_len=150_000, ploidy=2, snp=0.004, seed=909,
                                len_sd=2500.0, len_min=800, len_max=15000), 600)]
    for cfg, m in cases:
        seqs = [s for _, s in synth.oriented(synth.generate_reads(cfg))]
        want, secs, nrows = ck.reference_overlaps(seqs, m)
        assert nrows == len(want) > 1000
        got, st = hip_rows(seqs, m)
        same(got, want)
        got3, _ = hip_rows(seqs, m, shard=3)
        same(got3, want)


def test_many_identical_and_nested_reads():
    """Chains of many reads behind one prefix K-mer: 60 copies of one read, 40 of another that is a prefix of the
    first, plus suffix / infix variants -- every ordered pair overlaps or contains, four rows per identical pair
    (SURVEY.md section 8a-2).  Narrow and wide index, whole-set and sharded, with and without reverse strands."""
    rng = np.random.default_rng(123)
    base = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=1500))
    reads = [base] * 60 + [base[:900]] * 40 + [base[300:]] * 25 + [base[200:1100]] * 25 + [base[:64]] * 10
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    both = []
    for r in reads[::3]:
        both += [r, r.translate(rc)[::-1]]
    for seqs in (reads, both):
        for m in (64, 500):
            want = ck.oracle_overlaps(seqs, m)
            assert len(want) > 5000
            got, st = hip_rows(seqs, m)
            same(got, want, (len(seqs), m, "whole"))
            got, _ = hip_rows(seqs, m, shard=4)
            same(got, want, (len(seqs), m, "4 shards"))


@pytest.mark.parametrize("n_b,strands", [(100, 1), (100, 2), (240, 1), (240, 2)])
def test_tiles_with_more_survivors_than_one_probe_round(n_b, strands, capfd, monkeypatch):
    """One long read and n_b shorter ones that start at consecutive bases inside one 2048-base scan tile of it: that
    tile has n_b TRUE filter survivors -- more than the 64 one probe round settles (second round: 65..128), and with
    240 more than the two rounds together (the rest goes through the leftover list)."""
    monkeypatch.setenv("PHASM_DEBUG_LEFT", "1")
    rng = np.random.default_rng(77 + n_b)
    g = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=9000))
    reads = [g[:8000]] + [g[2100 + i:2100 + i + 1500 + (i % 7)] for i in range(n_b)] + [g[5000:9000]]
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for r in reads:
        seqs += [r] if strands == 1 else [r, r.translate(rc)[::-1]]
    want = ck.oracle_overlaps(seqs, 1000)
    assert len(want) > n_b * n_b // 4
    got, st = hip_rows(seqs, 1000)
    assert st["wide_index"] == 0
    same(got, want)
    got, _ = hip_rows(seqs, 1000, shard=3)
    same(got, want)
    deferred = [int(ln.split("deferred")[1].split()[0]) for ln in capfd.readouterr().err.splitlines() if ln.startswith("[left]")]
    if n_b > 128:
        assert max(deferred) >= n_b - 128      # survivors 129.. of that tile took the leftover path


def test_many_identical_reads_wide_index(monkeypatch):
    monkeypatch.setenv("PHASM_INDEX", "wide")
    rng = np.random.default_rng(321)
    base = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=1200))
    seqs = [base] * 50 + [base[:700]] * 30 + [base[400:]] * 30
    want = ck.oracle_overlaps(seqs, 100)
    got, st = hip_rows(seqs, 100)
    assert st["wide_index"] == 1
    same(got, want)
    got, _ = hip_rows(seqs, 100, shard=3)
    same(got, want)


def test_store1_rebuilt_on_the_device_is_what_the_host_packed(monkeypatch):
    """Reads added as (x, revcomp x) pairs: only the even reads cross PCIe, the odd ones are rebuilt on the device
    (k_revcomp_store).  PHASM_VERIFY_GENERATED uploads the host's own packed odd reads next to the rebuilt ones and
    fails the call on any differing word; PHASM_FULL_UPLOAD switches the shortcut off.  Same rows every way."""
    monkeypatch.setenv("PHASM_VERIFY_GENERATED", "1")
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    rng = np.random.default_rng(99)
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 100, 257, 2047, 2048, 2049, 4097]
    reads = []
    for ln in lens * 3:
        r = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=ln))
        reads += [r, r.translate(rc)[::-1]]
    ov = ExactOverlapper()
    for i, s in enumerate(reads):
        ov.add_sequence("r%d" % i, s)
    got = oo.sort_rows(oo.struct_to_rows(ov.overlaps_array(1)))
    st = ov.stats()
    ov.close()
    packed = sum(((len(s) + 31) // 32 + 1) * 8 for s in reads)
    assert st["paired"] == 1 and st["upload_bytes"] < 0.6 * packed + 16 * len(reads) + 4096
    _last.update(seqs=reads, m=1)
    same(got, ck.oracle_overlaps(reads, 1))
    for name in ("ladder_varlen", "cfg2_1k"):
        _, seqs, m, want = gu.ladder_case(name)
        got, st = hip_rows(seqs, m)
        assert st["paired"] == 1
        same(got, want, name)
    # one base off in one odd read: no shortcut, the pair check fails, still the right rows
    seqs = list(seqs)
    s = bytearray(seqs[41])
    s[7] = ord("A") if s[7] != ord("A") else ord("C")
    seqs[41] = bytes(s)
    got, st = hip_rows(seqs, m)
    assert st["paired"] == 0
    same(got, ck.oracle_overlaps(seqs, m))
    monkeypatch.delenv("PHASM_VERIFY_GENERATED")
    monkeypatch.setenv("PHASM_FULL_UPLOAD", "1")
    _, seqs, m, want = gu.ladder_case("cfg2_1k")
    got, st2 = hip_rows(seqs, m)
    assert st2["paired"] == 1 and st2["upload_bytes"] > 1.9 * sum(((len(s) + 31) // 32) * 8 for s in seqs[::2])
    same(got, want)


@pytest.mark.parametrize("chunks", [1, 2, 3, 4])
def test_rows_to_host_pipelined(chunks, monkeypatch):
    """po_overlaps_to_host: chunks of a-side reads, chunk k's rows copied to the host while chunk k + 1 is computed.
    Same rows as the goldens whatever the chunk count; the result serves rows(), repeated calls reuse the buffers."""
    monkeypatch.setenv("PHASM_HOST_CHUNKS", str(chunks))
    for name in ("ladder_varlen", "cfg2_1k", "cfg3_1k"):
        _, seqs, m, want = gu.ladder_case(name)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        _last.update(seqs=seqs, m=m)
        for rep in range(3):
            res = ov.overlaps_to_host_result(m if rep < 2 else 2 * m)
            arr = res.rows_view()
            assert len(arr) == len(res) == ov.stats()["n_rows"]
            got = oo.sort_rows(oo.struct_to_rows(arr))
            res.free()
            if rep < 2:
                same(got, want, "%s, %d chunks, call %d" % (name, chunks, rep))
            else:
                _last.update(seqs=seqs, m=2 * m)
                same(got, ck.oracle_overlaps(seqs, 2 * m), "%s at 2m" % name)
                _last.update(seqs=seqs, m=m)
        # an empty answer and the reference API's list of tuples come through the same path
        res = ov.overlaps_to_host_result(10_000_000)
        assert len(res) == 0 and len(res.rows()) == 0
        res.free()
        assert len(ov.overlaps(m)) == len(want)
        ov.close()


@pytest.mark.parametrize("n_slices", [2, 3, 8])
def test_sliced_wide_index_equals_the_whole_index(n_slices, monkeypatch):
    """Multi-GPU index: rank g builds sub-table g (keys partitioned by hash), the chunks are gathered, the shard calls
    probe the gathered index.  One GPU plays every rank in turn: N sub-tables exported into one buffer (the layout
    the all-gather produces), then the N shard calls on it -- the union must be the goldens' rows, also for a read
    set whose repeats crowd one sub-table."""
    import torch
    monkeypatch.setenv("PHASM_INDEX", "wide")
    cases = [gu.ladder_case("cfg3_1k")[1:], gu.ladder_case("ladder_varlen")[1:]] + [c[1:] for c in gu.repeats_cases()[2:]]
    # skew: read sets with a handful of distinct K-mers -- every key of the index lands in one or two sub-tables, whose
    # chain segments then hold (nearly) all entries while the other slices stay empty (the chunk is sized by the fullest)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    rng = np.random.default_rng(8)
    one = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=260))
    unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=7))
    period = [(unit * 60)[k:k + 300] for k in range(0, 28, 2)]
    for reads in ([one] * 40, period):
        seqs = [x for r in reads for x in (r, r.translate(rc)[::-1])]
        cases.append((seqs, 70, ck.oracle_overlaps(seqs, 70)))
    for seqs, m, want in cases:
        ov = ExactOverlapper(device=0)
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        built = [ov.index_slice_build(m, k, n_slices) for k in range(n_slices)]
        assert all(w for w, _, _ in built) and len({b for _, b, _ in built}) == 1
        bits, cap = built[0][1], max(e for _, _, e in built)
        chunk = ov.index_chunk_bytes(bits, cap)
        buf = torch.empty(n_slices * chunk, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for k in range(n_slices):
            w, b, e = ov.index_slice_build(m, k, n_slices)
            assert (w, b, e) == built[k]
            ov.index_slice_export(buf.data_ptr() + k * chunk, cap)
        parts = [ov.overlaps_shard_indexed_array(m, k, n_slices, buf.data_ptr(), n_slices, bits, cap) for k in range(n_slices)]
        got = oo.sort_rows(oo.struct_to_rows(np.concatenate(parts)))
        _last.update(seqs=seqs, m=m)
        same(got, want, "%d slices" % n_slices)
        # the candidate form (what travels between the GPUs) on the same index, expanded
        res, written = ov.candidates_result_indexed(m, 0, n_slices, buf.data_ptr(), n_slices, bits, cap)
        assert not written and ov.stats()["wide_index"] == 1
        res.free()
        # and the ordinary call afterwards builds its own whole index again
        same(oo.sort_rows(oo.struct_to_rows(ov.overlaps_result(m).rows())), want, "whole index after the slices")
        ov.close()


def test_row_ranges_of_a_result():
    _, seqs, m, want = gu.ladder_case("cfg2_1k")
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    for res in (ov.overlaps_result(m), ov.overlaps_to_host_result(m)):
        n = len(res)
        whole = res.rows()
        cuts = [0, 1, n // 3, n // 2, n - 1, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            assert np.array_equal(res.rows_range_view(lo, hi - lo), whole[lo:hi])
        assert len(res.rows_range_view(n, 0)) == 0
        with pytest.raises(RuntimeError):
            res.rows_range_view(n - 1, 2)
        res.free()
    ov.close()


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_sharded_upload_pieces_assemble_to_the_same_read_set(nshards, monkeypatch):
    """Multi-GPU upload: rank g uploads the words of shard g's even reads, the pieces are gathered, po_upload_assemble
    puts them in place and rebuilds the odd reads.  One GPU plays every rank; PHASM_VERIFY_GENERATED compares the
    rebuilt odd store with the host's word by word (a misplaced piece cannot pass), and the rows are the goldens'."""
    import torch
    monkeypatch.setenv("PHASM_VERIFY_GENERATED", "1")
    for name in ("ladder_varlen", "cfg2_1k"):
        _, seqs, m, want = gu.ladder_case(name)
        ov = ExactOverlapper(device=0)
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        sizes = [ov.upload_piece(k, nshards) for k in range(nshards)]
        assert all(ok for ok, _ in sizes)
        slot = max(n for _, n in sizes) + 1
        buf = torch.zeros(nshards * slot, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for k in range(nshards):
            ok, n = ov.upload_piece(k, nshards, buf.data_ptr() + 8 * k * slot, slot)
            assert ok and n == sizes[k][1]
        ov.upload_assemble(buf.data_ptr(), slot, nshards)
        _last.update(seqs=seqs, m=m)
        same(oo.sort_rows(oo.struct_to_rows(ov.overlaps_result(m).rows())), want, "%s, %d pieces" % (name, nshards))
        assert ov.stats()["paired"] == 1 and ov.stats()["upload_bytes"] < 8 * (sizes[0][1] + slot) + 16 * len(seqs) + 4096
        # the same in parts (the H2D of part k + 1 runs under the all-gather of part k): gathered layout [part][shard][slot]
        for parts in (2, 5):
            psz = [[ov.upload_piece_part(k, nshards, q, parts) for q in range(parts)] for k in range(nshards)]
            assert all(sum(n for _, n in row) == sizes[k][1] for k, row in enumerate(psz))    # the parts tile the piece
            pslot = max(n for row in psz for _, n in row) + 1
            pbuf = torch.full((parts * nshards * pslot,), -1, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            ov.invalidate()
            for q in range(parts):
                for k in range(nshards):
                    ok, n = ov.upload_piece_part(k, nshards, q, parts, pbuf.data_ptr() + 8 * (q * nshards + k) * pslot, pslot)
                    assert ok and n == psz[k][q][1]
            ov.upload_assemble(pbuf.data_ptr(), pslot, nshards, parts)
            same(oo.sort_rows(oo.struct_to_rows(ov.overlaps_result(m).rows())), want, "%s, %d pieces x %d parts" % (name, nshards, parts))
        ov.close()
    # reads that are not strand pairs: no sharded upload
    ov = ExactOverlapper(device=0)
    ov.add_sequence("a", "ACGTACGTTGCA" * 10)
    ov.add_sequence("b", "TTGCAACGTAGG" * 10)
    assert ov.upload_piece(0, 2) == (False, 0)
    ov.close()


def test_rows_to_host_on_unpaired_odd_and_8bit_read_sets(monkeypatch):
    monkeypatch.setenv("PHASM_HOST_CHUNKS", "3")
    rng = np.random.default_rng(31337)
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=5000))
    reads = [genome[int(s):int(s) + int(l)] for s, l in zip(rng.integers(0, 4000, size=41), rng.integers(60, 900, size=41))]
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    g8 = alpha[rng.integers(0, len(alpha), size=4000)].tobytes()
    reads8 = [g8[int(s):int(s) + int(l)] for s, l in zip(rng.integers(0, 3000, size=37), rng.integers(40, 700, size=37))]
    for seqs, m, bits in ((reads, 40, 2), (reads8, 20, 8)):
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        res = ov.overlaps_to_host_result(m)
        got = oo.sort_rows(oo.struct_to_rows(res.rows()))
        res.free()
        st = ov.stats()
        ov.close()
        assert st["bits_per_base"] == bits and st["paired"] == 0
        _last.update(seqs=seqs, m=m)
        same(got, ck.oracle_overlaps(seqs, m), "%d-bit, %d reads, 3 chunks" % (bits, len(seqs)))
