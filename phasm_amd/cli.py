"""``phasm overlap`` on the MI355X path.

Counterpart of ``overlap(args)`` in the reference CLI (/root/reference/phasm/cli/assembler.py:29-50,
parser at :436-452): same positional FASTA argument, ``-l/--min-length`` (default 1000),
``-o/--output`` (default stdout), and byte-identical ``H`` / ``S`` / ``E`` lines for identical
rows -- all ``S`` lines first, then the ``E`` lines.  Row order differs from the reference
(whose order is an artefact of ``std::unordered_map`` iteration, overlapper.cpp:30,:68): here
rows are a-major in FASTA order, then by start position.

    python -m phasm_amd.cli overlap reads.fasta -l 1000 -o overlaps.gfa
"""
from __future__ import annotations

import argparse
import logging
import sys

from .io import gfa
from .io.fasta import read_fasta, reverse_complement
from .overlapper import ExactOverlapper

logger = logging.getLogger("phasm_amd")


def overlap(args) -> int:
    args.output.write(gfa.gfa_header())
    overlapper = ExactOverlapper(device=getattr(args, "device", None))
    logger.info("Packing reads and searching for pairwise overlaps on the GPU...")
    if isinstance(args.fasta_input, (str, bytes)) and not getattr(args, "python_ingest", False):
        # native parse + reverse complement + 2-bit pack in one pass (po_add_fasta)
        overlapper.add_fasta(args.fasta_input, both_strands=True)
        ids, lens = overlapper.ids(), overlapper.lengths()
        for i in range(0, len(ids), 2):
            args.output.write(gfa.gfa_line("S", ids[i][:-1], int(lens[i]), "*"))
    else:
        for name, seq in read_fasta(args.fasta_input):
            args.output.write(gfa.gfa_line("S", name, len(seq), "*"))
            overlapper.add_sequence(name + "+", seq)
            overlapper.add_sequence(name + "-", reverse_complement(seq))
    res = overlapper.overlaps_result(args.min_length)
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
    logger.info("Done.")
    return n


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
    p.add_argument("fasta_input", help="FASTA file with reads")
    p.set_defaults(func=overlap)
    args = parser.parse_args(argv)
    if not getattr(args, "func", None):
        parser.print_help()
        return 1
    logging.basicConfig(level=[logging.WARNING, logging.INFO, logging.DEBUG][min(args.verbose, 2)],
                        stream=sys.stderr)
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
