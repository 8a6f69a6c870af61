"""Loaders for the committed golden vectors (tests/golden/, made by make_golden.py from the
compiled reference).  Each case is ``(name, [seq bytes...], min_length, expected_rows)`` with
expected_rows a lexicographically sorted (n,6) int64 array of
``(a_idx, b_idx, astart, aend, bstart, bend)``."""
import hashlib
import json
import os

import numpy as np

from phasm_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rows(x):
    return np.asarray(x, dtype=np.int64).reshape(-1, 6)


def sha(seqs):
    h = hashlib.sha256()
    for s in seqs:
        h.update(len(s).to_bytes(4, "little"))
        h.update(s)
    return h.hexdigest()


def json_cases(fname):
    with open(os.path.join(GOLDEN, fname)) as f:
        cases = json.load(f)
    return [(c["name"], [s.encode("latin-1") for s in c["reads"]], c["min_length"], _rows(c["rows"]))
            for c in cases]


def repeats_cases():
    z = np.load(os.path.join(GOLDEN, "repeats.npz"))
    cat = z["cat"].tobytes()
    lens = z["lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    seqs = [cat[offs[i]:offs[i + 1]] for i in range(len(lens))]
    assert sha(seqs) == str(z["sha256"])
    return [("repeats_m%d" % m, seqs, int(m), _rows(z["rows_m%d" % m])) for m in z["min_lengths"]]


LADDER_NAMES = ["ladder_small", "ladder_varlen", "ladder_cfg1_mini", "ladder_cfg2_mini",
                "ladder_cfg4_noise", "ladder_lownoise", "cfg1_full", "cfg2_1k", "cfg3_1k", "cfg5_1k"]


def ladder_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = synth.SynthConfig(**json.loads(str(z["config"])))
    seqs = [s for _, s in synth.oriented(synth.generate_reads(cfg))]
    # the generator is seeded; the hash proves these are the reads the reference saw
    assert sha(seqs) == str(z["sha256"]), "synthetic generator drifted from the golden inputs"
    return name, seqs, int(z["min_length"]), _rows(z["rows"])


def all_small_cases():
    return json_cases("toy_cases.json") + json_cases("adversarial.json")
