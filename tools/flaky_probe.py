#!/usr/bin/env python3
"""Developer probe: replay one trial of the seeded fuzz (tests/test_gpu_parity.py) many times with dirty device
memory in between, and print how the rows differ from the oracle's when they do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle import overlap_oracle as oo  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402


def trials(seed, n):
    rng = np.random.default_rng(seed)
    rc = bytes.maketrans(b"ACGT", b"TGCA")
    for trial in range(n):
        glen = int(rng.integers(300, 6000))
        if rng.random() < 0.3:
            unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(1, 12))))
            genome = bytearray((unit * (glen // len(unit) + 1))[:glen])
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
                reads.append(r)
        if rng.random() < 0.6:
            seqs = []
            for r in reads:
                seqs += [r, r.translate(rc)[::-1]]
        else:
            seqs = reads
        m = int(rng.choice([1, 2, 5, 31, 32, 33, 62, 63, 64, 100, 500]))
        yield trial, seqs, m


def hip(seqs, m):
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    arr = ov.overlaps_array(m)
    st = ov.stats()
    ov.close()
    return oo.sort_rows(oo.struct_to_rows(arr)), st


if __name__ == "__main__":
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    os.environ["PHASM_INDEX"] = sys.argv[3] if len(sys.argv) > 3 else "wide"
    for trial, seqs, m in trials(2024, which + 1):
        pass
    want = oo.oracle_overlaps(seqs, m)
    print("trial", trial, "reads", len(seqs), "m", m, "rows", len(want), flush=True)
    bad = 0
    for r in range(reps):
        # dirty the allocator's free memory: whatever the library allocates next is not zero
        junk = torch.full((96 << 20,), 0xAB, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
        got, st = hip(seqs, m)
        if not np.array_equal(got, want):
            bad += 1
            from collections import Counter
            cg, cw = Counter(map(tuple, got.tolist())), Counter(map(tuple, want.tolist()))
            extra, missing = list((cg - cw).elements()), list((cw - cg).elements())
            print("rep", r, "rows", len(got), "extra", extra[:12], "missing", missing[:12], {k: st[k] for k in ("n_candidates", "n_verified", "n_rows", "wide_index", "paired")}, flush=True)
    print("mismatching runs:", bad, "of", reps)
