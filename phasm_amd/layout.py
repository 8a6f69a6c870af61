"""Host mirror of stage 1 of ``phasm layout`` -- the consumer of the overlap rows.

The reference (/root/reference/phasm/cli/assembler.py:52-139) reads the GFA2 file twice (segments,
then ``E`` lines), turns every line into a ``LocalAlignment`` (phasm/io/gfa.py:90-104), pushes it
through ``ContainedReads`` / ``MinReadLength`` / ``MinOverlapLength`` / ``MaxOverhang``
(phasm/filter.py:37-122), feeds the survivors to ``build_assembly_graph``
(phasm/assembly_graph.py:136-179) and finally deletes every filtered read in both orientations
(assembler.py:108-126).  Here the same result -- the edge set of the graph at "Final graph"
(assembler.py:136) -- comes from ``po_layout_edges`` on the device, either straight from the rows of
``po_overlaps`` (no file in between) or from a GFA2 file read natively (``po_add_gfa``).

Graph cleaning after that point (transitive reduction, tips, bubbles, merging) is out of scope.
No CPU fallback: without the HIP library and a GPU these functions raise.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from .overlapper import ExactOverlapper, OverlapResult

# `phasm layout` defaults, assembler.py:469-489
DEFAULTS = dict(min_read_length=0, min_overlap_length=0, max_overhang_abs=1000, max_overhang_rel=0.8)


@dataclass
class AssemblyEdges:
    """Edges of the assembly graph after stage 1: ``edges`` is a structured array (u, v, weight,
    overlap_len) with u, v oriented-read indices into ``ids``; ``contained[i]`` tells that read i (nodes 2i,
    2i+1) was contained in another read and left the graph."""
    edges: np.ndarray
    contained: np.ndarray
    ids: List[str]
    stats: dict

    def edge_tuples(self) -> List[Tuple[str, str, int, int]]:
        ids = self.ids
        e = self.edges
        return [(ids[u], ids[v], w, o) for u, v, w, o in
                zip(e["u"].tolist(), e["v"].tolist(), e["weight"].tolist(), e["overlap_len"].tolist())]

    def to_networkx(self):
        """A ``networkx.DiGraph`` with the reference's edge attributes (``weight``, ``overlap_len``)."""
        import networkx
        g = networkx.DiGraph()
        for u, v, w, o in self.edge_tuples():
            g.add_edge(u, v, weight=w, overlap_len=o)
        return g


def build_assembly_graph(ov: ExactOverlapper, rows: OverlapResult, min_read_length: int = 0,
                         min_overlap_length: int = 0, max_overhang_abs: int = 1000,
                         max_overhang_rel: float = 0.8) -> AssemblyEdges:
    """Filters + ``build_assembly_graph`` + contained-read removal on a row result of ``ov``."""
    res, removed = ov.layout_edges(rows, min_read_length, min_overlap_length, max_overhang_abs, max_overhang_rel)
    try:
        edges = res.rows()
    finally:
        res.free()
    return AssemblyEdges(edges, removed.astype(bool), ov.ids(), ov.layout_stats())


def layout_from_gfa(path: str, device: Optional[int] = None, **params) -> AssemblyEdges:
    """``phasm layout`` stage 1 from an overlap file: native GFA2 read, then the device passes."""
    ov = ExactOverlapper(device=device)
    try:
        _, rows = ov.add_gfa(path)
        try:
            return build_assembly_graph(ov, rows, **{**DEFAULTS, **params})
        finally:
            rows.free()
    finally:
        ov.close()


def load_daligner(ov: ExactOverlapper, db_input, las_input, translations=None) -> OverlapResult:
    """DBdump + LAdump text -> the reads (as segments) and the row result on ``ov``, i.e. the state after
    ``daligner2gfa`` and ``po_add_gfa`` without the file in between (phasm_amd/io/daligner.py ``to_rows``)."""
    from .io import daligner
    names, lengths, rows = daligner.to_rows(db_input, las_input, translations)
    for name, n in zip(names, lengths.tolist()):
        ov.add_segment(name, n)
    return ov.result_from_rows(rows)


def layout_from_daligner(db_input, las_input, translations=None, device: Optional[int] = None, **params) -> AssemblyEdges:
    """``phasm layout`` stage 1 straight from DAZZ_DB / DALIGNER dump text."""
    ov = ExactOverlapper(device=device)
    try:
        rows = load_daligner(ov, db_input, las_input, translations)
        try:
            return build_assembly_graph(ov, rows, **{**DEFAULTS, **params})
        finally:
            rows.free()
    finally:
        ov.close()


def layout_from_overlaps(ov: ExactOverlapper, min_length: int, **params) -> AssemblyEdges:
    """Overlap + layout stage 1 without the file in between: the rows never leave HBM."""
    rows = ov.overlaps_result(min_length)
    try:
        return build_assembly_graph(ov, rows, **{**DEFAULTS, **params})
    finally:
        rows.free()
