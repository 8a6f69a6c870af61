#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the *reference overlapper itself*.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every expected row in the fixtures is an output of oracle/_ref/ref_overlapper, i.e. of
/root/reference/src/overlapper.cpp compiled in place (oracle/Makefile).  Only data is
written here: inputs (explicit reads, or generator parameters + sha256 of the generated
reads) and the reference's rows as sorted index tuples.  No reference source is stored.

Fixtures
  toy_cases.json      hand-made known-answer cases (SURVEY.md section 8c table + edge cases)
  adversarial.json    seeded random small cases: tiny alphabets, substrings of a common
                      genome, duplicated reads, empty reads, non-ACGT / lower-case bytes
  repeats.npz         low-complexity reads (tandem repeats, homopolymers) -> exercises the
                      "longest only" rule and many occurrences per pair
  ladder_*.npz        seeded synthetic read sets (phasm_amd.synth), both strands added as the
                      CLI does (assembler.py:38-40)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import overlap_oracle as oo  # noqa: E402
from phasm_amd import synth  # noqa: E402


def ref_rows(seqs, m):
    rows, _, _ = oo.reference_overlaps(seqs, m)
    return rows


def sha(seqs):
    h = hashlib.sha256()
    for s in seqs:
        h.update(len(s).to_bytes(4, "little"))
        h.update(s)
    return h.hexdigest()


def toy_cases():
    cases = [
        ("chain", ["AAACCCGGGTTT", "GGGTTTACGTAC", "CCCGGG"], 3),
        ("suffix_contained_dup", ["AAACCCGGG", "CCCGGG"], 3),
        ("identical_reads", ["ACGTTGCA", "ACGTTGCA"], 3),
        ("longest_only", ["TGGGGGG", "GGGGGGTA"], 3),
        ("prefix_contained", ["ACGTACGGTT", "ACGTAC"], 3),
        ("every_occurrence", ["TTACGTTACGTT", "ACGT"], 3),
        ("min_len_inclusive_3", ["TTTTACG", "ACGCCCC"], 3),
        ("min_len_inclusive_4", ["TTTTACG", "ACGCCCC"], 4),
        ("palindrome_strands", ["AAACCCGGGTTT", "AAACCCGGGTTT", "GGGTTTACGTAC", "GTACGTAAACCC"], 4),
        ("min_len_zero", ["TTACG", "ACGCC"], 0),
        ("min_len_one", ["TTACG", "ACGCC"], 1),
        ("empty_read_present", ["ACGT", "", "ACGT"], 0),
        ("single_read", ["ACGTACGT"], 2),
        ("short_reads_below_min", ["ACG", "CGT", "ACGTACGTAC", "GTACGGGG"], 4),
        ("homopolymer", ["AAAAAAAA", "AAAAA", "AAAAAAAAAAAA"], 3),
        ("lowercase_is_distinct", ["ACGTacgt", "acgtTTTT", "ACGTTTTT"], 3),
        ("n_bases", ["ACGNNNAC", "NNNACGGT", "NNACG"], 3),
        ("three_identical", ["GATTACA", "GATTACA", "GATTACA"], 2),
        ("periodic_prefix", ["CACACACACA", "ACACACAT", "CACACA"], 2),
        ("k_boundary_33", ["T" + "ACGT" * 10, "ACGT" * 10 + "GG"], 33),
        ("self_overlap_only", ["ACGTTTACGT"], 3),
    ]
    out = []
    for name, seqs, m in cases:
        rows = ref_rows(seqs, m)
        out.append({"name": name, "reads": seqs, "min_length": m, "rows": rows.tolist()})
    return out


def adversarial(n_trials=400, seed=12345):
    rng = np.random.default_rng(seed)
    out = []
    alphabets = [b"AC", b"ACG", b"ACGT", b"ACGT", b"ACGTN", b"ACGTacgt"]
    for t in range(n_trials):
        alpha = alphabets[rng.integers(len(alphabets))]
        glen = int(rng.integers(8, 80))
        genome = bytes(alpha[i] for i in rng.integers(0, len(alpha), size=glen))
        nreads = int(rng.integers(2, 10))
        reads = []
        for _ in range(nreads):
            mode = rng.integers(0, 10)
            if mode == 0 and reads:
                reads.append(reads[rng.integers(len(reads))])          # duplicate read
            elif mode == 1:
                ln = int(rng.integers(0, 12))                          # random (maybe empty)
                reads.append(bytes(alpha[i] for i in rng.integers(0, len(alpha), size=ln)))
            else:
                ln = int(rng.integers(1, min(40, glen) + 1))           # substring of genome
                st = int(rng.integers(0, glen - ln + 1))
                reads.append(genome[st:st + ln])
        m = int(rng.integers(0, 7))
        seqs = [r.decode("latin-1") for r in reads]
        rows = ref_rows(seqs, m)
        out.append({"name": "adv%d" % t, "reads": seqs, "min_length": m, "rows": rows.tolist()})
    return out


def pack_reads(seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    return cat, lens


def repeats_fixture(seed=777):
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(40):
        kind = rng.integers(0, 4)
        if kind == 0:
            parts.append(b"ACGT"[rng.integers(4):][:1] * int(rng.integers(30, 150)))
        elif kind == 1:
            unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(2, 9))))
            parts.append(unit * int(rng.integers(10, 40)))
        else:
            parts.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(20, 120)))))
    genome = b"".join(parts)
    reads = []
    for _ in range(120):
        ln = int(rng.integers(40, 400))
        st = int(rng.integers(0, len(genome) - ln))
        r = genome[st:st + ln]
        if rng.random() < 0.5:
            r = r.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        reads.append(r)
    seqs = []
    for r in reads:  # both strands, as the CLI adds them
        seqs.append(r)
        seqs.append(r.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])
    res = {}
    for m in (20, 33, 64):
        res["rows_m%d" % m] = ref_rows(seqs, m).astype(np.int32)
    cat, lens = pack_reads(seqs)
    np.savez_compressed(os.path.join(HERE, "repeats.npz"), cat=cat, lens=lens,
                        min_lengths=np.array([20, 33, 64]), sha256=np.array(sha(seqs)), **res)
    print("repeats.npz", {k: len(v) for k, v in res.items()})


LADDER = {
    # name: (SynthConfig, min_length)
    "ladder_small": (synth.SynthConfig(n_reads=80, read_len=2000, genome_len=10_000, ploidy=2,
                                       snp=0.005, seed=11), 200),
    "ladder_varlen": (synth.SynthConfig(n_reads=150, read_len=1500, genome_len=12_000, ploidy=3,
                                        snp=0.004, seed=12, len_sd=600.0, len_min=100,
                                        len_max=4000), 100),
    "ladder_cfg1_mini": (synth.SynthConfig(n_reads=200, read_len=10_000, genome_len=100_000,
                                           ploidy=2, snp=0.005, seed=1, len_sd=1500.0), 1000),
    "ladder_cfg2_mini": (synth.SynthConfig(n_reads=300, read_len=15_000, genome_len=60_000,
                                           ploidy=2, snp=0.005, seed=2), 1000),
    "ladder_cfg4_noise": (synth.SynthConfig(n_reads=120, read_len=15_000, genome_len=40_000,
                                            ploidy=2, snp=0.005, seed=4, noise=0.01), 1000),
    # low noise: some overlaps survive it, so the noise branch of synth.expected_rows is pinned by non-empty output
    "ladder_lownoise": (synth.SynthConfig(n_reads=200, read_len=3000, genome_len=20_000, ploidy=2,
                                          snp=0.004, seed=14, noise=0.0004), 300),
    # BASELINE.json configs[0] at FULL size: 1k reads (~10 kb), min-overlap 1000 -- the reference's own
    # CPU-runnable case (about 40 s of reference time)
    "cfg1_full": (synth.CONFIGS["cfg1"], 1000),
    # BASELINE.json configs[1] density (75x per haplotype) at 1 000 reads = 2 000 oriented x 15 kb
    "cfg2_1k": (synth.scaled(synth.CONFIGS["cfg2"], 1000), 1000),
    # BASELINE.json configs[2] density (triploid, 50x per haplotype) and configs[4] density (tetraploid, 12 kb
    # reads, 30x per haplotype) at 1 000 reads = 2 000 oriented reads each
    "cfg3_1k": (synth.scaled(synth.CONFIGS["cfg3"], 1000), 1000),
    "cfg5_1k": (synth.scaled(synth.CONFIGS["cfg5"], 1000), 1000),
}


def ladder_reads(cfg):
    return [s for _, s in synth.oriented(synth.generate_reads(cfg))]


def ladder(only=None):
    for name, (cfg, m) in LADDER.items():
        if only and name not in only:
            continue
        seqs = ladder_reads(cfg)
        rows, secs, _ = oo.reference_overlaps(seqs, m)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            config=np.array(json.dumps(cfg.describe())), min_length=np.array(m),
                            sha256=np.array(sha(seqs)), n_oriented=np.array(len(seqs)),
                            rows=rows.astype(np.int32), ref_seconds=np.array(secs))
        print(name, "oriented reads", len(seqs), "rows", len(rows), "ref %.1fs" % secs)


def main():
    if not oo.have_reference():
        sys.exit("oracle/_ref/ref_overlapper missing: run `make -C oracle ref` first")
    if len(sys.argv) > 1:   # only the named ladder cases (the others are left as committed)
        ladder(set(sys.argv[1:]))
        return
    with open(os.path.join(HERE, "toy_cases.json"), "w") as f:
        json.dump(toy_cases(), f, indent=0)
    adv = adversarial()
    with open(os.path.join(HERE, "adversarial.json"), "w") as f:
        json.dump(adv, f, separators=(",", ":"))
    print("adversarial rows:", sum(len(c["rows"]) for c in adv))
    repeats_fixture()
    ladder()


if __name__ == "__main__":
    main()
