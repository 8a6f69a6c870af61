#!/usr/bin/env python3
"""Generate tests/golden/daligner_cases.json from the REFERENCE's DBdump / LAdump parsers.

Runs only in the build container (needs /root/reference); the output is committed.  For every case the
reference's own ``phasm.io.daligner.parse_reads`` / ``parse_local_alignments`` parse the two dump texts and the
records they yield are stored (numpy byte strings decoded).  The expected GFA2 text is assembled with the
reference's ``phasm.io.gfa.gfa_header`` / ``gfa_line`` from those records in the way ``daligner2gfa`` does
(/root/reference/phasm/cli/convert.py:65-133).  ``phasm.cli.convert`` itself cannot be imported here: it
imports dinopy at module level, which this image does not have -- so the forty lines of that command are
followed by ``reference_gfa`` below rather than executed, and with ``with_sequences`` the bases are written
(the reference prints the repr of a 0-d numpy array there).  Cases where the reference raises store the
exception's class name.
"""
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from phasm.io import daligner, gfa  # noqa: E402  (reference)
from phasm.io.daligner import Strand  # noqa: E402


def reference_gfa(db, las, with_sequences, spacing, translations):
    out = [gfa.gfa_header(trace_spacing=spacing)]
    internal, lengths = {}, {}
    for read in daligner.parse_reads(io.StringIO(db)):
        rid = read["read_id"]
        if translations:
            rid = translations[daligner.full_id(read)]
        rid = rid.split()[0]
        internal[read["read_id"]] = rid
        lengths[read["read_id"]] = read["length"]
        seq = read["sequence"].item().decode("ascii") if with_sequences and "sequence" in read else "*"
        out.append(gfa.gfa_line("S", rid, str(read["length"]), seq))
    for la in daligner.parse_local_alignments(io.StringIO(las)):
        parts = ["E", "*", internal[la["a"]] + "+",
                 internal[la["b"]] + ("+" if la["strand"] == Strand.SAME else "-")]
        ar, br = list(map(str, la["arange"])), list(map(str, la["brange"]))
        if la["arange"][1] == lengths[la["a"]]:
            ar[1] += "$"
        if la["brange"][1] == lengths[la["b"]]:
            br[1] += "$"
        parts.extend(ar + br)
        parts.append(",".join(str(t[1]) for t in la["trace_points"]) if spacing and "trace_points" in la else "*")
        out.append(gfa.gfa_line(*parts))
    return "".join(out)


def plain(rec):
    r = dict(rec)
    if "sequence" in r:
        r["sequence"] = r["sequence"].item().decode("ascii")
    if "strand" in r:
        r["strand"] = int(r["strand"])
    for k in ("arange", "brange"):
        if k in r:
            r[k] = list(r[k])
    if "trace_points" in r:
        r["trace_points"] = [list(t) for t in r["trace_points"]]
    return r


def attempt(fn):
    try:
        return {"ok": fn()}
    except Exception as e:      # noqa: BLE001 -- the class name is the datum
        return {"raises": type(e).__name__}


def dump_db(rng, n, with_seq, movie):
    lengths, lines, pos = [], [], 0
    lines += ["+ R %d" % n, "+ M 1", "+ H %d" % (len(movie) * n), "@ H %d" % len(movie)]
    for i in range(n):
        l = rng.randint(30, 900)
        start = rng.choice([0, 0, rng.randint(1, 5000)])
        lengths.append(l)
        lines.append("R %d" % (i + 1))
        lines.append("H %d %s" % (len(movie), movie))
        lines.append("L %d %d %d" % (rng.randint(0, 160000), start, start + l))
        if with_seq:
            lines.append("S %d %s" % (l, "".join(rng.choice("acgt") for _ in range(l))))
    return "\n".join(lines) + "\n", lengths


def dump_las(rng, lengths, n_la, trace, spacing=100):
    n = len(lengths)
    lines = ["+ P %d" % n_la, "%% P %d" % n_la, "+ T 9", "@ T 9"]
    for _ in range(n_la):
        a, b = rng.randrange(n), rng.randrange(n)
        la, lb = lengths[a], lengths[b]
        kind = rng.random()
        if kind < 0.5:
            l = rng.randint(10, min(la, lb))
            s, e, bs, be = la - l, la, 0, l
        elif kind < 0.7:
            l = rng.randint(10, min(la, lb))
            s, e, bs, be = 0, l, lb - l, lb
        else:
            s = rng.randint(0, la - 1)
            e = rng.randint(s + 1, la)
            bs = rng.randint(0, lb - 1)
            be = rng.randint(bs + 1, lb)
        lines.append("P %d %d %s %s" % (a + 1, b + 1, rng.choice("nc"), rng.choice("o.-+")))
        lines.append("C %d %d %d %d" % (s, e, bs, be))
        if trace:
            k = max(1, (e - s) // spacing + 1)
            lines.append("T %d" % k)
            for _ in range(k):
                lines.append("   %d %d" % (rng.randint(0, 12), rng.randint(60, 130)))
        if rng.random() < 0.7:
            lines.append("D %d" % rng.randint(0, 40))
    return "\n".join(lines) + "\n"


def main():
    cases = []

    def add(name, db, las, with_sequences=False, spacing=None, translations=None):
        cases.append({"name": name, "db": db, "las": las, "with_sequences": with_sequences, "spacing": spacing,
                      "translations": translations,
                      "reads": attempt(lambda: [plain(r) for r in daligner.parse_reads(io.StringIO(db))]),
                      "alignments": attempt(lambda: [plain(r) for r in daligner.parse_local_alignments(io.StringIO(las))]),
                      "gfa": attempt(lambda: reference_gfa(db, las, with_sequences, spacing, translations))})

    for seed in range(24):
        rng = random.Random(4000 + seed)
        movie = "m%06d_c%d" % (seed, rng.randint(1, 10**12))
        with_seq = seed % 3 == 0
        trace = seed % 2 == 0
        db, lengths = dump_db(rng, rng.randint(2, 14), with_seq, movie)
        las = dump_las(rng, lengths, rng.randint(1, 60), trace)
        trans = None
        if seed % 4 == 1:     # fasta2dazzdb's name map: full id -> original name, some with a description after a blank
            trans = {}
            for r in daligner.parse_reads(io.StringIO(db)):
                trans[daligner.full_id(r)] = "orig_%s%s" % (r["read_id"], " some description" if int(r["read_id"]) % 2 else "")
        if seed % 8 == 5:     # two dump reads mapped to one external name
            ks = sorted(trans)
            trans[ks[-1]] = trans[ks[0]]
        add("random_%d" % seed, db, las, with_sequences=with_seq and seed % 6 == 0,
            spacing=(100 if trace and seed % 4 == 0 else None), translations=trans)

    db2 = "R 1\nH 3 mov\nL 0 0 8\nR 2\nH 3 mov\nL 1 5 9\n"
    add("c_before_first_p", db2, "C 9 9 9 9\nP 1 2 n o\nP 2 1 c o\nC 1 4 0 3\n")
    add("two_c_lines", db2, "P 1 2 n o\nC 0 1 0 1\nC 4 8 0 4\n")
    add("trace_without_spacing", db2, "P 1 2 n o\nC 4 8 0 4\nT 2\n   1 2\n   0 2\n")
    add("trace_empty_list", db2, "P 1 2 c o\nC 4 8 0 4\nT 1\nD 3\n", spacing=50)
    add("no_alignments", db2, "+ P 0\n% P 0\n")
    add("no_reads", "+ R 0\n", "")
    add("windows_newlines", db2.replace("\n", "\r\n"), "P 1 2 n o\r\nC 4 8 0 4\r\n")
    add("extra_blanks", "R   1\nH 3   mov\nL  0  0  8\nR 2\nH 3 mov\nL 1 5 9\n", "P  1  2  n  o\nC  4 8  0 4\n")
    add("read_without_r_first", "H 3 mov\nL 0 0 8\nR 2\nH 3 mov\nL 1 5 9\n", "")
    # malformed input: the class of the exception is the expectation
    add("bad_r_fields", "R 1 2\n", "")
    add("bad_h_fields", "R 1\nH mov\n", "")
    add("bad_l_fields", "R 1\nH 3 mov\nL 0 8\n", "")
    add("bad_l_number", "R 1\nH 3 mov\nL 0 x 8\n", "")
    add("s_without_bases", "R 1\nH 3 mov\nL 0 0 8\nS 8\n", "")
    add("read_without_length", "R 1\nH 3 mov\n", "")
    add("p_too_short", db2, "P 1 2\n")
    add("c_too_short", db2, "P 1 2 n o\nC 1 2 3\n")
    add("c_too_long", db2, "P 1 2 n o\nC 1 2 3 4 5\n")
    add("p_without_c", db2, "P 1 2 n o\nP 2 1 n o\nC 0 1 0 1\n")
    add("unknown_read", db2, "P 1 7 n o\nC 0 1 0 1\n")
    add("trace_points_after_t0", db2, "P 1 2 n o\nC 4 8 0 4\nT 0\n   1 2\n")
    add("trace_points_without_t", db2, "P 1 2 n o\nC 4 8 0 4\n   1 2\n")
    add("short_trace_line", db2, "P 1 2 n o\nC 4 8 0 4\nT 1\n   7\n", spacing=100)
    add("missing_translation", db2, "P 1 2 n o\nC 4 8 0 4\n", translations={"mov/1/0_8": "x"})

    hashes = {n: daligner.generate_moviename_hash(n) for n in ["reads.fasta", "a b.fq", "", "ünï.fa", "x" * 300]}

    class R:                      # what fix_header needs of a dinopy read record
        def __init__(self, name, seq):
            self.name, self.sequence = name, seq
    mapping = {}
    fixed = [(s.decode(), n.decode()) for s, n in
             daligner.fix_header([R(b"first read", b"ACGT"), R(b"second", b"AC" * 40)], "12345", mapping)]
    path = os.path.join(HERE, "daligner_cases.json")
    with open(path, "w") as f:
        json.dump({"cases": cases, "moviename_hash": hashes, "fix_header": {"moviename": "12345", "out": fixed, "map": mapping}},
                  f, indent=0)
    ok = sum(1 for c in cases if "ok" in c["gfa"])
    print("wrote", path, len(cases), "cases (%d convert, %d raise)" % (ok, len(cases) - ok), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
