import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper
d = tempfile.mkdtemp(dir="/tmp")
fa = os.path.join(d, "reads.fasta")
synth.write_fasta(fa, synth.generate_reads(synth.CONFIGS["cfg2"]))
for it in range(2):
    t0 = time.time(); ov = ExactOverlapper(); ov.add_fasta(fa); t1 = time.time()
    ids = ov.ids(); lens = ov.lengths(); t2 = time.time()
    res = ov.overlaps_result(1000); t3 = time.time()
    with open(os.path.join(d, "o.gfa"), "w") as f:
        n = res.write_gfa_edges(f)
    t4 = time.time()
    res.free(); ov.close()
    print("add_fasta %.2f  ids/lengths %.2f  overlaps(+upload) %.2f  write %.2f" % (t1-t0, t2-t1, t3-t2, t4-t3))
