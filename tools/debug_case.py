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
    arr = ov.overlaps_array(m)
    st = ov.stats()
    ov.close()
    return oo.sort_rows(oo.struct_to_rows(arr)), st

for name, seqs, m, want in gu.json_cases("toy_cases.json")[:8] + [gu.ladder_case("ladder_small")]:
    got, st = run(seqs, m)
    ok = np.array_equal(got, want)
    print(name, "OK" if ok else "DIFF", "got", len(got), "want", len(want), "cand", st["n_candidates"], "tiles", st["n_tiles"], "paired", st["paired"])
    if not ok and len(want) < 20:
        print(" got:", got.tolist()); print(" want:", want.tolist())

name, seqs, m, want = gu.ladder_case("ladder_small")
for w in ("16", "4", "1"):
    os.environ["PHASM_SCAN_WAVES"] = w
    for rep in range(3):
        got, st = run(seqs, m)
        print("waves", w, "rep", rep, "rows", len(got), "cand", st["n_candidates"])
