import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import golden_utils as gu
from oracle import overlap_oracle as oo
from phasm_amd.overlapper import ExactOverlapper
def run(seqs, m):
    ov = ExactOverlapper()
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    arr = ov.overlaps_array(m); st = ov.stats(); ov.close()
    return oo.sort_rows(oo.struct_to_rows(arr)), st
bad = 0
for name, seqs, m, want in gu.all_small_cases():
    got, st = run(seqs, m)
    if not np.array_equal(got, want):
        bad += 1
        if bad <= 4:
            print(name, "m", m, "bits", st["bits_per_base"], "K", st["kmer"], "cand", st["n_candidates"], "reads", [s.decode() for s in seqs])
            print("  got ", got.tolist()); print("  want", want.tolist())
print("bad", bad)
