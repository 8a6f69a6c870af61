"""``phasm overlap`` on the MI355X path.

Counterpart of ``overlap(args)`` in the reference CLI (/root/reference/phasm/cli/assembler.py:29-50,
parser at :436-452): same positional FASTA argument, ``-l/--min-length`` (default 1000),
``-o/--output`` (default stdout), and byte-identical ``H`` / ``S`` / ``E`` lines for identical
rows -- all ``S`` lines first, then the ``E`` lines.  Row order differs from the reference
(whose order is an artefact of ``std::unordered_map`` iteration, overlapper.cpp:30,:68): here
rows are a-major in FASTA order, then by start position.

    python -m phasm_amd.cli overlap reads.fasta -l 1000 -o overlaps.gfa

``layout-edges`` is the first stage of ``phasm layout`` (assembler.py:52-139, options :469-489) on the
same device: read the overlap file, classify and filter the alignments, drop contained reads, and
write the assembly graph as it stands before graph cleaning, in the reference's own graph format
(``gfa2_write_graph``, phasm/io/gfa.py:283-327: one ``S`` line per read that still has an edge, one
``E * u v weight len(u) 0 overlap_len *`` line per edge).

    python -m phasm_amd.cli layout-edges overlaps.gfa -o graph.gfa

``daligner2gfa`` is the other producer of the overlap file (``phasm-convert daligner2gfa``,
/root/reference/phasm/cli/convert.py:65-133, options :146-183): DBdump + LAdump text in, the same GFA2 lines out.
``layout-edges --las`` takes the two dumps directly and skips the text in between.

    python -m phasm_amd.cli daligner2gfa -o overlaps.gfa reads.dbdump alignments.ladump
    python -m phasm_amd.cli layout-edges reads.dbdump --las alignments.ladump -o graph.gfa
"""
from __future__ import annotations

import argparse
import logging
import sys

from .io import daligner, gfa
from .io.fasta import read_fasta, reverse_complement
from .overlapper import ExactOverlapper
from . import layout as layout_mod

logger = logging.getLogger("phasm_amd")


def _seconds_since_process_start() -> float:
    """Wall time since the kernel started this process (interpreter start and imports included), from /proc."""
    import os
    try:
        with open("/proc/self/stat") as f:
            start_ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return float("nan")


def overlap(args) -> int:
    import time
    marks = [("process start -> overlap() entered (interpreter, imports)", _seconds_since_process_start())]
    t_prev = [time.perf_counter()]

    def mark(what):
        now = time.perf_counter()
        marks.append((what, now - t_prev[0]))
        t_prev[0] = now

    args.output.write(gfa.gfa_header())
    overlapper = ExactOverlapper(device=getattr(args, "device", None))
    mark("library loaded, handle made")
    logger.info("Packing reads and searching for pairwise overlaps on the GPU...")
    if isinstance(args.fasta_input, (str, bytes)) and not getattr(args, "python_ingest", False):
        # native parse + reverse complement + 2-bit pack in one pass (po_add_fasta)
        overlapper.add_fasta(args.fasta_input, both_strands=True)
        mark("FASTA parsed, reverse-complemented, packed (the device comes up beside it)")
        try:
            args.output.fileno()
            overlapper.write_gfa_segments(args.output)       # S lines formatted in C, straight to the descriptor
        except (AttributeError, OSError, ValueError):
            ids, lens = overlapper.ids(), overlapper.lengths()
            for i in range(0, len(ids), 2):
                args.output.write(gfa.gfa_line("S", ids[i][:-1], int(lens[i]), "*"))
    else:
        for name, seq in read_fasta(args.fasta_input):
            args.output.write(gfa.gfa_line("S", name, len(seq), "*"))
            overlapper.add_sequence(name + "+", seq)
            overlapper.add_sequence(name + "-", reverse_complement(seq))
    mark("S lines")
    max_diff = int(getattr(args, "max_diff", 0) or 0)
    if max_diff > 0:
        # beyond the reference (which is exact, assembler.py:436-439): banded seed-extension DP, po_overlaps_ex
        res = overlapper.overlaps_ex_result(args.min_length, max_diff, int(getattr(args, "band", 0) or 0))
    else:
        res = overlapper.overlaps_to_host_result(args.min_length)   # rows travel to the host while later chunks are computed
    mark("the overlap call (upload, kernels, rows home)")
    logger.info("Writing %d overlaps to GFA2...", len(res))
    try:
        try:
            args.output.fileno()
            native = True
        except (AttributeError, OSError, ValueError):
            native = False
        if native:
            n = res.write_gfa_edges(args.output)       # formatted in C, straight to the descriptor
        else:
            n = gfa.write_edges(args.output, res.rows(), overlapper.ids())
    finally:
        res.free()
    mark("E lines formatted and written")
    args.output.flush()
    if not getattr(args, "leave_handle_to_exit", False):
        overlapper.close()
    mark("flush + handle closed")
    if getattr(args, "timing", False):
        import json
        sys.stderr.write("PHASM_CLI_TIMING " + json.dumps({k: round(v, 4) for k, v in marks}) + "\n")
    logger.info("Done.")
    return n


def write_stage1_graph(out, ids, lengths, edges_arr, edges_res=None) -> int:
    """The graph after stage 1 in the format of gfa2_write_graph (phasm/io/gfa.py:283-327): S lines of the reads
    that still have an edge, then one ``E * u v weight len(u) 0 overlap_len *`` line per edge (written natively
    from the device result when ``out`` is a real file)."""
    import numpy as np
    out.write(gfa.gfa_header())
    e = edges_arr
    used = np.zeros(len(ids) // 2, dtype=bool)
    used[e["u"] >> 1] = True
    used[e["v"] >> 1] = True
    for i in np.flatnonzero(used).tolist():
        out.write(gfa.gfa_line("S", ids[2 * i][:-1], int(lengths[2 * i]), "*"))
    if edges_res is not None:
        try:
            out.fileno()
            return edges_res.write_gfa_edges(out)
        except (AttributeError, OSError, ValueError):
            pass
    lu = np.asarray(lengths)[e["u"]]
    chunk = 1 << 18
    for lo in range(0, len(e), chunk):
        sl = slice(lo, lo + chunk)
        out.write("".join("E\t*\t%s\t%s\t%d\t%d\t0\t%d\t*\n" % t for t in
                          zip([ids[u] for u in e["u"][sl].tolist()], [ids[v] for v in e["v"][sl].tolist()],
                              e["weight"][sl].tolist(), lu[sl].tolist(), e["overlap_len"][sl].tolist())))
    return len(e)


def _load_translations(path):
    """-T of daligner2gfa (convert.py:69-76): a JSON object {dump id: name}; a missing file is ignored with a
    warning (the reference's warning line itself fails on an undefined attribute, :75-76)."""
    import json
    import os
    if not path:
        return None
    if not os.path.isfile(path):
        logger.warning("Translations file '%s' does not exist, ignoring.", path)
        return None
    with open(path) as f:
        return json.load(f)


def daligner2gfa(args) -> int:
    las = sys.stdin if args.las_input in (None, "-") else open(args.las_input)
    try:
        with open(args.db_input) as db:
            n_s, n_e = daligner.write_gfa(args.out, db, las, args.with_sequences, args.with_trace_points,
                                          _load_translations(args.translations))
    finally:
        if las is not sys.stdin:
            las.close()
    logger.info("Wrote %d reads and %d local alignments.", n_s, n_e)
    return n_e


def layout_edges(args) -> int:
    ov = ExactOverlapper(device=getattr(args, "device", None))
    try:
        if getattr(args, "las", None):
            with open(args.gfa_file) as db, open(args.las) as las:
                rows = layout_mod.load_daligner(ov, db, las, _load_translations(getattr(args, "translations", None)))
            nseg = len(ov) // 2
        else:
            nseg, rows = ov.add_gfa(args.gfa_file)
        logger.info("Read %d reads and %d local alignments from the GFA2 file.", nseg, len(rows))
        try:
            edges, _removed = ov.layout_edges(rows, args.min_read_length, args.min_overlap_length,
                                              args.max_overhang_abs, args.max_overhang_rel)
        finally:
            rows.free()
        st = ov.layout_stats()
        logger.info("%d contained reads removed; %d alignments pass the filters; graph has %d edges.",
                    st["n_contained_reads"], st["n_pass"], st["n_edges"])
        try:
            return write_stage1_graph(args.output, ov.ids(), ov.lengths(), edges.rows(), edges)
        finally:
            edges.free()
    finally:
        ov.close()


def main(argv=None) -> int:
    parser = argparse.ArgumentParser(prog="phasm-amd", description="MI355X-native PHASM overlap step")
    parser.add_argument("-v", "--verbose", action="count", default=0)
    sub = parser.add_subparsers(dest="command")
    p = sub.add_parser("overlap", help="Find pairwise exact overlaps between reads in a FASTA file.")
    p.add_argument("-l", "--min-length", type=int, default=1000,
                   help="Minimum overlap length (default: 1000)")
    p.add_argument("-o", "--output", type=argparse.FileType("w"), default=sys.stdout,
                   help="Output file (default: stdout)")
    p.add_argument("--device", type=int, default=None, help="HIP device ordinal (default 0)")
    p.add_argument("--python-ingest", action="store_true", help="parse the FASTA in Python instead of po_add_fasta")
    p.add_argument("--timing", action="store_true", help="print the command's stage times as one JSON line on stderr")
    p.add_argument("--leave-handle-to-exit", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--max-diff", type=int, default=0,
                   help="(extension beyond the exact reference) accept overlaps with up to this many differences: banded "
                        "seed-extension DP on the GPU; 0 = exact, the reference's behaviour (default)")
    p.add_argument("--band", type=int, default=8, help="with --max-diff: diagonals each side of the anchor's diagonal (<= 30; default 8)")
    p.add_argument("fasta_input", help="FASTA file with reads")
    p.set_defaults(func=overlap)
    # option names and defaults of `phasm layout`, assembler.py:469-489
    q = sub.add_parser("layout-edges", help="Stage 1 of `phasm layout`: filter the alignments of an overlap file and "
                                            "write the assembly graph before graph cleaning.")
    q.add_argument("-l", "--min-read-length", type=int, default=0)
    q.add_argument("-s", "--min-overlap-length", type=int, default=0)
    q.add_argument("-a", "--max-overhang-abs", type=int, default=1000)
    q.add_argument("-r", "--max-overhang-rel", type=float, default=0.8)
    q.add_argument("-o", "--output", type=argparse.FileType("w"), default=sys.stdout)
    q.add_argument("--device", type=int, default=None)
    q.add_argument("--las", default=None, metavar="LADUMP",
                   help="read the positional file as DBdump text and the alignments from this LAdump text")
    q.add_argument("-T", "--translations", default=None, help="with --las: JSON map of dump ids to read names")
    q.add_argument("gfa_file", help="GFA2 file with the reads (S lines) and their pairwise alignments (E lines)")
    q.set_defaults(func=layout_edges)
    # option names of `phasm-convert daligner2gfa`, convert.py:146-183
    c = sub.add_parser("daligner2gfa", help="Convert DAZZ_DB DBdump and DALIGNER LAdump text to a GFA2 overlap file.")
    c.add_argument("-s", "--with-sequences", action="store_true", default=False)
    c.add_argument("-t", "--with-trace-points", type=int, default=None, metavar="SPACING")
    c.add_argument("-T", "--translations", type=str, default=None, metavar="TRANSLATION_FILE")
    c.add_argument("-o", "--out", type=argparse.FileType("w"), default=sys.stdout)
    c.add_argument("db_input", help="DBdump text")
    c.add_argument("las_input", nargs="?", default=None, help="LAdump text (default: stdin)")
    c.set_defaults(func=daligner2gfa)
    args = parser.parse_args(argv)
    if not getattr(args, "func", None):
        parser.print_help()
        return 1
    logging.basicConfig(level=[logging.WARNING, logging.INFO, logging.DEBUG][min(args.verbose, 2)],
                        stream=sys.stderr)
    args.func(args)
    for name in ("output", "out"):   # (the command may end with os._exit: nothing may be left in a Python buffer)
        f = getattr(args, name, None)
        if f is not None and f not in (sys.stdout, sys.stderr):
            f.close()
        elif f is not None:
            f.flush()
    return 0


def _run_as_command() -> None:
    """The command ends when its output is on disk: the handle's device buffers, the page-locked pools and the GPU runtime
    are left to the operating system (giving 600 MB of page-locked and device memory back buffer by buffer, and unloading
    the runtime, took 0.15 s of the `overlap` command's 0.8 s)."""
    import os
    argv = sys.argv[1:]
    if argv and argv[0] == "overlap" and "--keep-teardown" not in argv:
        argv = ["overlap", "--leave-handle-to-exit"] + argv[1:]
    else:
        argv = [a for a in argv if a != "--keep-teardown"]
    rc = 1
    try:
        rc = main(argv)
        sys.stdout.flush()
        sys.stderr.flush()
    except SystemExit as e:
        raise e
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    os._exit(rc)


if __name__ == "__main__":
    _run_as_command()
