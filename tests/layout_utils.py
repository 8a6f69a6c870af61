"""Loader for tests/golden/layout_cases.json (made by tests/golden/make_layout_golden.py from the
reference's own filter / alignment classes)."""
import hashlib
import json
import os
import random

import numpy as np

import golden_utils
from phasm_amd.io import gfa

GOLDEN = golden_utils.GOLDEN


def digest(xs):
    return hashlib.sha256(json.dumps([int(x) for x in xs]).encode()).hexdigest()


def gfa_text(names, lengths, rows):
    out = ["H\tVN:z:2.0\n"]
    for n, l in zip(names, lengths):
        out.append("S\t%s\t%d\t*\n" % (n, l))
    for a, b, s, e, bs, be in rows:
        out.append("E\t*\t%s%s\t%s%s\t%d\t%d\t%d\t%d\t*\n" % (names[a >> 1], "+-"[a & 1], names[b >> 1], "+-"[b & 1], s, e, bs, be))
    return "".join(out)


def load_cases():
    """-> list of dict(name, params, names, lengths, rows (n,6) int64, expect, text)."""
    with open(os.path.join(GOLDEN, "layout_cases.json")) as f:
        raw = json.load(f)
    out = []
    for c in raw:
        if "gfa" in c:
            text = c["gfa"]
            names, lengths, rows = gfa.read_gfa2_rows(text.splitlines(True))
        else:
            # rows of a committed overlap golden, in stored or seeded-shuffled order (as the generator did)
            _, seqs, _, grows = golden_utils.ladder_case(c["ladder"])
            names = ["read%d" % i for i in range(len(seqs) // 2)]
            lengths = np.array([len(seqs[2 * i]) for i in range(len(names))], dtype=np.int64)
            rl = [tuple(int(x) for x in r) for r in grows]
            if c["shuffle_seed"] is not None:
                random.Random(c["shuffle_seed"]).shuffle(rl)
            rows = np.array(rl, dtype=np.int64).reshape(-1, 6)
            text = gfa_text(names, lengths.tolist(), rl)
        assert len(rows) == c["n_rows"]
        out.append(dict(name=c["name"], params=c["params"], names=names, lengths=np.asarray(lengths), rows=rows,
                        expect=c["expect"], text=text, digests="gfa" not in c))
    return out


def node_lengths(lengths):
    return np.repeat(np.asarray(lengths, dtype=np.int64), 2)


def node_name(names, node):
    return names[node >> 1] + "+-"[node & 1]
