#!/usr/bin/env python3
"""Developer probe: overlaps of a synthetic config, then po_layout_edges on the rows still in HBM."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--min-length", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    if a.reads:
        cfg = synth.scaled(cfg, a.reads)
    ov = ExactOverlapper()
    for name, seq in synth.oriented(synth.generate_reads(cfg)):
        ov.add_sequence(name, seq)
    res = ov.overlaps_result(a.min_length)
    print("rows", len(res), file=sys.stderr)
    for it in range(a.iters):
        t0 = time.time()
        edges, _ = ov.layout_edges(res, want_removed=False)
        dt = time.time() - t0
        st = ov.layout_stats()
        edges.free()
        st["wall_ms"] = dt * 1e3
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
    res.free()
    ov.close()
