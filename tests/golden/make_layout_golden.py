#!/usr/bin/env python3
"""Generate tests/golden/layout_cases.json from the REFERENCE's own Python classes.

Runs only in the build container (needs /root/reference); the output is committed.  For each case a
GFA2 text (S + E lines) is pushed through exactly what `phasm layout` does before
``build_assembly_graph`` (/root/reference/phasm/cli/assembler.py:56-100):

    reads   = phasm.io.gfa.gfa2_parse_segments(file)
    la_iter = map(phasm.io.gfa.gfa2_line_to_la(reads), E lines)
    filters = [ContainedReads(), MinReadLength(n)?, MinOverlapLength(n)?, MaxOverhang(abs, rel)]
    filter(lambda x: all(f(x) for f in filters), la_iter)

and the reference objects' answers are recorded: per line ``classify()``, ``get_overlap_length()``,
``get_overhang()``; which lines reached the graph builder; per filter ``filtered`` and
``nodes_to_remove``.  ``build_assembly_graph`` itself cannot run under the installed networkx 3.4 (it
uses the 1.x ``add_edge(u, v, attr_dict)`` signature), so no edge attributes are recorded here.

Case sources: (1) the rows of the committed overlap goldens (outputs of the compiled reference
overlapper), in stored and in shuffled order; (2) seeded random alignments that reach every branch
(all four types, non-zero bstart, `$` positions, both strands, repeated read pairs, short reads,
thresholds where ``ratio * overlap`` is not an integer); (3) GFA text in the form the reference's other producer
of this wire format writes (daligner2gfa: TS header tag, `$` ends, trace-point lists, b on either strand).
"""
import hashlib
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")

from phasm.filter import ContainedReads, MaxOverhang, MinOverlapLength, MinReadLength  # noqa: E402  (reference)
from phasm.io import gfa  # noqa: E402  (reference)

import golden_utils  # noqa: E402


def run_reference(text, params):
    reads = gfa.gfa2_parse_segments(io.StringIO(text))
    filters = [ContainedReads()]
    if params["min_read_length"]:
        filters.append(MinReadLength(params["min_read_length"]))
    if params["min_overlap_length"]:
        filters.append(MinOverlapLength(params["min_overlap_length"]))
    filters.append(MaxOverhang(params["max_overhang_abs"], params["max_overhang_rel"]))
    mapper = gfa.gfa2_line_to_la(reads)
    types, ovl, hang, passed = [], [], [], []
    k = 0
    for line in io.StringIO(text):
        if not line.startswith("E"):
            continue
        la = mapper(line)
        types.append(int(la.classify()))
        ovl.append(int(la.get_overlap_length()))
        hang.append(int(la.get_overhang()))
        if all(f(la) for f in filters):
            passed.append(k)
        k += 1
    return {"n_segments": len(reads), "types": types, "overlap_len": ovl, "overhang": hang, "passed": passed,
            "filters": [{"name": f.__class__.__name__, "filtered": f.filtered,
                         "nodes_to_remove": sorted(str(r) for r in f.nodes_to_remove)} for f in filters]}


def digest(xs):
    return hashlib.sha256(json.dumps(xs).encode()).hexdigest()


def gfa_text(names, lengths, rows, dollar=False):
    out = ["H\tVN:z:2.0\n"]
    for n, l in zip(names, lengths):
        out.append("S\t%s\t%d\t*\n" % (n, l))
    for a, b, s, e, bs, be in rows:
        la, lb = lengths[a >> 1], lengths[b >> 1]
        es = "%d$" % e if dollar and e == la else "%d" % e
        bes = "%d$" % be if dollar and be == lb else "%d" % be
        out.append("E\t*\t%s%s\t%s%s\t%d\t%s\t%d\t%s\t*\n" % (names[a >> 1], "+-"[a & 1], names[b >> 1], "+-"[b & 1], s, es, bs, bes))
    return "".join(out)


def overlap_case(name, seqs, rows, params, order_seed=None):
    """Rows of an overlap golden (reads were added as x+, x-: node index = row index)."""
    assert len(seqs) % 2 == 0
    names = ["read%d" % i for i in range(len(seqs) // 2)]
    lengths = [len(seqs[2 * i]) for i in range(len(names))]
    assert all(len(seqs[2 * i + 1]) == lengths[i] for i in range(len(names)))
    rows = [tuple(int(x) for x in r) for r in rows]
    if order_seed is not None:
        random.Random(order_seed).shuffle(rows)
    return name, names, lengths, rows, params


def random_case(seed):
    rng = random.Random(seed)
    n = rng.randint(3, 12)
    names = ["r%d_%d" % (seed, i) if i % 3 else "read %d/%d" % (seed, i) for i in range(n)]   # a name with a blank
    lengths = [rng.choice([rng.randint(20, 60), rng.randint(60, 400)]) for _ in range(n)]
    rows = []
    for _ in range(rng.randint(5, 60)):
        a = rng.randrange(2 * n)
        b = rng.randrange(2 * n)
        la, lb = lengths[a >> 1], lengths[b >> 1]
        kind = rng.random()
        if kind < 0.35:      # proper dovetail a -> b with small overhangs
            l = rng.randint(5, min(la, lb))
            oh1, oh2 = rng.choice([0, 0, 1, 3, 10]), rng.choice([0, 0, 2, 7])
            s = la - l - oh2 if la - l - oh2 >= 0 else 0
            row = (a, b, s, min(la, s + l), min(oh1, lb), min(lb, oh1 + l))
        elif kind < 0.55:    # dovetail b -> a
            l = rng.randint(5, min(la, lb))
            oh = rng.choice([0, 0, 2, 9])
            s = lb - l - oh if lb - l - oh >= 0 else 0
            row = (a, b, min(oh, la), min(la, oh + l), s, min(lb, s + l))
        elif kind < 0.75:    # containment either way
            if la <= lb:
                p = rng.randint(0, lb - la)
                row = (a, b, 0, la, p, p + la)
            else:
                p = rng.randint(0, la - lb)
                row = (a, b, p, p + lb, 0, lb)
        else:                # anything
            s = rng.randint(0, la - 1)
            e = rng.randint(s + 1, la)
            bs = rng.randint(0, lb - 1)
            be = rng.randint(bs + 1, lb)
            row = (a, b, s, e, bs, be)
        rows.append(row)
        if rng.random() < 0.2:   # the same ordered pair again with other coordinates (add_edge overwrites)
            s = rng.randint(0, la - 1)
            rows.append((a, b, s, rng.randint(s + 1, la), 0, rng.randint(1, lb)))
    params = {"min_read_length": rng.choice([0, 0, 30, 70]), "min_overlap_length": rng.choice([0, 0, 8, 25]),
              "max_overhang_abs": rng.choice([1000, 5, 12, 0]), "max_overhang_rel": rng.choice([0.8, 0.8, 0.35, 0.07, 1.5])}
    return "random_%d" % seed, names, lengths, rows, params


def daligner_form_case(seed):
    """E lines the way the other producer of this wire format writes them (daligner2gfa,
    /root/reference/phasm/cli/convert.py:104-131): a always on '+', b on either strand, a `$` after an end
    position that equals the read length, a comma-separated trace-point list in the alignment field, and the
    TS tag in the header (:78).  Returned as ready GFA text."""
    rng = random.Random(seed)
    n = rng.randint(4, 10)
    names = ["m%d/%d/0_%d" % (seed, i, 100 + i) for i in range(n)]
    lengths = [rng.randint(80, 600) for _ in range(n)]
    out = ["H\tVN:z:2.0\tTS:i:100\n"]
    for nm, l in zip(names, lengths):
        out.append("S\t%s\t%d\t*\n" % (nm, l))
    n_rows = 0
    for _ in range(rng.randint(10, 50)):
        a, b = rng.randrange(n), rng.randrange(n)
        la, lb = lengths[a], lengths[b]
        if rng.random() < 0.6:      # dovetail, a's end on b's start, a few bases of overhang
            l = rng.randint(20, min(la, lb))
            oh = rng.choice([0, 0, 1, 4, 15])
            s, e, bs, be = max(la - l - oh, 0), la - oh if la - oh > 0 else la, min(oh, lb - 1), min(lb, oh + l)
            if e <= s:
                e = la
            if be <= bs:
                be = lb
        else:
            s = rng.randint(0, la - 1)
            e = rng.randint(s + 1, la)
            bs = rng.randint(0, lb - 1)
            be = rng.randint(bs + 1, lb)
        es = "%d$" % e if e == la else "%d" % e
        bes = "%d$" % be if be == lb else "%d" % be
        tp = ",".join(str(rng.randint(0, 9)) for _ in range(rng.randint(1, 6)))
        out.append("E\t*\t%s+\t%s%s\t%d\t%s\t%d\t%s\t%s\n" % (names[a], names[b], rng.choice("+-"), s, es, bs, bes, tp))
        n_rows += 1
    params = {"min_read_length": rng.choice([0, 150]), "min_overlap_length": rng.choice([0, 40]),
              "max_overhang_abs": rng.choice([1000, 10]), "max_overhang_rel": rng.choice([0.8, 0.2])}
    return "daligner_form_%d" % seed, "".join(out), n_rows, params


def main():
    default = {"min_read_length": 0, "min_overlap_length": 0, "max_overhang_abs": 1000, "max_overhang_rel": 0.8}
    cases = []
    toys = golden_utils.json_cases("toy_cases.json")
    for name, seqs, m, rows in toys:
        if len(seqs) % 2 == 0 and len(seqs) >= 2 and all(len(seqs[2 * i]) == len(seqs[2 * i + 1]) for i in range(len(seqs) // 2)):
            cases.append(("toy_" + name,) + overlap_case(name, seqs, rows, default)[1:])
    for lad, params, seed in [("ladder_small", default, None), ("ladder_varlen", default, None),
                              ("ladder_varlen", dict(default, min_read_length=4000, min_overlap_length=300), 7),
                              ("ladder_cfg1_mini", default, None),
                              ("ladder_cfg1_mini", dict(default, min_read_length=9000, max_overhang_abs=0), 11)]:
        name, seqs, m, rows = golden_utils.ladder_case(lad)
        tag = lad + ("" if seed is None else "_shuffled%d" % seed)
        cases.append((tag,) + overlap_case(lad, seqs, rows, params, seed)[1:] + (lad, seed))
    for seed in range(40):
        cases.append(random_case(1000 + seed))

    out = []
    for c in cases:
        name, names, lengths, rows, params = c[:5]
        text = gfa_text(names, lengths, rows, dollar=name.startswith("random") and int(name.split("_")[1]) % 2 == 0)
        exp = run_reference(text, params)
        if len(c) > 5:
            # big cases: rows come from the overlap golden at test time; keep digests of the long lists
            ladder, seed = c[5], c[6]
            hist = [exp["types"].count(t) for t in range(4)]
            out.append({"name": name, "ladder": ladder, "shuffle_seed": seed, "params": params, "n_rows": len(rows),
                        "expect": {"n_segments": exp["n_segments"], "type_hist": hist,
                                   "types_sha256": digest(exp["types"]), "overlap_len_sha256": digest(exp["overlap_len"]),
                                   "overhang_sha256": digest(exp["overhang"]), "n_passed": len(exp["passed"]),
                                   "passed_sha256": digest(exp["passed"]), "filters": exp["filters"]}})
        else:
            out.append({"name": name, "gfa": text, "params": params, "n_rows": len(rows), "expect": exp})
    for seed in range(8):
        name, text, n_rows, params = daligner_form_case(2000 + seed)
        out.append({"name": name, "gfa": text, "params": params, "n_rows": n_rows, "expect": run_reference(text, params)})
    path = os.path.join(HERE, "layout_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(out), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
