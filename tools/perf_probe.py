#!/usr/bin/env python3
"""Developer probe: load a synthetic config, run po_overlaps a few times, print per-stage times."""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper


def load(cfg):
    t0 = time.time()
    reads = synth.generate_reads(cfg)
    t1 = time.time()
    ov = ExactOverlapper()
    for name, seq in synth.oriented(reads):
        ov.add_sequence(name, seq)
    t2 = time.time()
    ov.upload()
    t3 = time.time()
    print("gen %.1fs add %.1fs upload %.2fs" % (t1 - t0, t2 - t1, t3 - t2), file=sys.stderr)
    return ov


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--min-length", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--max-diff", type=int, default=-1, help=">= 0: po_overlaps_ex (banded DP) with this many differences")
    ap.add_argument("--band", type=int, default=0)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    if a.reads:
        cfg = synth.scaled(cfg, a.reads)
    ov = load(cfg)
    for it in range(a.iters):
        t0 = time.time()
        res = ov.overlaps_result(a.min_length) if a.max_diff < 0 else ov.overlaps_ex_result(a.min_length, a.max_diff, a.band)
        dt = time.time() - t0
        st = ov.stats()
        res.free()
        st["wall_ms"] = dt * 1e3
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
