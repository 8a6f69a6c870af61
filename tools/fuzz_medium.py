#!/usr/bin/env python3
"""One-off medium-size fuzz: a few hundred reads of up to 8 kb from diploid/triploid genomes with planted
repeats, both strands, random min_length; HIP rows (narrow / wide, whole / sharded) against the CPU oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import overlap_oracle as oo
from phasm_amd.overlapper import ExactOverlapper

def hip(seqs, m, shard=None):
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    arr = ov.overlaps_array(m) if shard is None else np.concatenate([ov.overlaps_shard_array(m, k, shard) for k in range(shard)])
    ov.close()
    return oo.sort_rows(oo.struct_to_rows(arr))

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
rc = bytes.maketrans(b"ACGT", b"TGCA")
for t in range(trials):
    glen = int(rng.integers(20_000, 120_000))
    g = bytearray(bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=glen)))
    for _ in range(int(rng.integers(0, 4))):            # planted repeats (copies of a segment, a tandem array)
        ln = int(rng.integers(200, 3000)); src = int(rng.integers(0, glen - ln)); dst = int(rng.integers(0, glen - ln))
        g[dst:dst + ln] = g[src:src + ln]
    if rng.random() < 0.4:
        unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(2, 40))))
        at = int(rng.integers(0, glen - 4000)); rep = (unit * (4000 // len(unit) + 1))[:4000]; g[at:at + 4000] = rep
    haps = [bytes(g)]
    for _ in range(int(rng.integers(1, 3))):
        h = bytearray(g)
        for pos in rng.integers(0, glen, size=int(glen * 0.004)):
            h[pos] = b"ACGT"[rng.integers(4)]
        haps.append(bytes(h))
    reads = []
    for _ in range(int(rng.integers(150, 500))):
        ln = int(rng.integers(300, 8000)); ln = min(ln, glen); st = int(rng.integers(0, glen - ln + 1))
        r = haps[int(rng.integers(len(haps)))][st:st + ln]
        if rng.random() < 0.5:
            r = r.translate(rc)[::-1]
        reads.append(r)
    seqs = []
    for r in reads:
        seqs += [r, r.translate(rc)[::-1]]
    m = int(rng.choice([20, 63, 64, 200, 1000, 2500]))
    want = oo.oracle_overlaps(seqs, m)
    for idx in ("narrow", "wide"):
        os.environ["PHASM_INDEX"] = idx
        for order in ("0", "1"):
            os.environ["PHASM_VERIFY_ORDER"] = order
            got = hip(seqs, m)
            assert np.array_equal(got, want), (seed, t, idx, order, "whole")
        got = hip(seqs, m, shard=5)
        assert np.array_equal(got, want), (seed, t, idx, "5 shards")
    print("trial %d ok: %d reads, m=%d, %d rows" % (t, len(seqs), m, len(want)), flush=True)
print("all ok")
