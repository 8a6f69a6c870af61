#!/usr/bin/env python3
"""Time the whole `overlap` command on a config-2-sized FASTA (developer probe)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth, cli
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
if len(sys.argv) > 2:
    cfg = synth.scaled(cfg, int(sys.argv[2]))
d = tempfile.mkdtemp(dir="/tmp")
fa = os.path.join(d, "reads.fasta")
t = time.time(); synth.write_fasta(fa, synth.generate_reads(cfg)); print("fasta written %.1fs (%d MB)" % (time.time() - t, os.path.getsize(fa) >> 20))
for extra in ([], ["--python-ingest"]):
    out = os.path.join(d, "out%d.gfa" % len(extra))
    t = time.time(); cli.main(["overlap", fa, "-l", "1000", "-o", out] + extra)
    print("overlap %s: %.2fs, output %d MB" % (extra, time.time() - t, os.path.getsize(out) >> 20))
# stage 1 of `phasm layout` on the file just written
import ctypes
from phasm_amd.overlapper import ExactOverlapper
out = os.path.join(d, "out0.gfa")
t = time.time(); ov = ExactOverlapper(); nseg, rows = ov.add_gfa(out); t1 = time.time() - t
t = time.time(); edges, removed = ov.layout_edges(rows); t2 = time.time() - t
st = ov.layout_stats()
print("layout: po_add_gfa %.2fs (%d segments, %d rows); po_layout_edges %.3fs (device %.2f ms, first call incl. H2D of the rows); %d edges, %d contained reads"
      % (t1, nseg, len(rows), t2, st["ms_total"], len(edges), st["n_contained_reads"]))
edges.free(); rows.free(); ov.close()
graph = os.path.join(d, "graph.gfa")
t = time.time(); cli.main(["layout-edges", out, "-o", graph])
print("layout-edges command: %.2fs, output %d MB" % (time.time() - t, os.path.getsize(graph) >> 20))
