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
