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
    ap.add_argument("--windows", default="", help="comma-separated PHASM_WIDE_WINDOW values: the resident call for each of them with the "
                                                 "index rebuilt every time (PHASM_NO_INDEX_REUSE): index build and scan passes per window")
    ap.add_argument("--host", action="store_true", help="time one host-to-host call: invalidate + upload + po_overlaps_to_host + rows")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    if a.reads:
        cfg = synth.scaled(cfg, a.reads)
    ov = load(cfg)
    if a.host:
        for it in range(a.iters + 1):
            t0 = time.perf_counter()
            ov.invalidate()
            ov.upload()
            t1 = time.perf_counter()
            res = ov.overlaps_to_host_result(a.min_length)
            n = len(res.rows_view())
            t2 = time.perf_counter()
            st = ov.stats()
            res.free()
            if it:
                print(json.dumps({"rows": n, "upload_ms": round((t1 - t0) * 1e3, 3), "upload_MB": round(st["upload_bytes"] / 1e6, 1),
                                  "kernels_plus_d2h_ms": round((t2 - t1) * 1e3, 3), "step_ms": round((t2 - t0) * 1e3, 3),
                                  "overlaps_per_sec": round(n / (t2 - t0)), "kernel_ms_sum": round(st["ms_total"], 3),
                                  "index_ms": round(st["ms_index"], 3), "wide": st["wide_index"]}))
        sys.exit(0)
    if a.windows:
        os.environ["PHASM_NO_INDEX_REUSE"] = "1"
        for w in a.windows.split(","):
            os.environ["PHASM_WIDE_WINDOW"] = w.strip()
            for it in range(a.iters):
                t0 = time.time()
                res = ov.overlaps_result(a.min_length)
                dt = time.time() - t0
                st = ov.stats()
                n = len(res)
                res.free()
                print(json.dumps({"window": int(w), "rows": n, "wall_ms": round(dt * 1e3, 3), "candidates": st["n_candidates"],
                                  **{k: round(st[k], 3) for k in ("ms_index", "ms_scan_count", "ms_scan_probe", "ms_scan_fill", "ms_verify", "ms_select", "ms_emit", "ms_total")}}), flush=True)
        sys.exit(0)
    for it in range(a.iters):
        t0 = time.time()
        res = ov.overlaps_result(a.min_length) if a.max_diff < 0 else ov.overlaps_ex_result(a.min_length, a.max_diff, a.band)
        dt = time.time() - t0
        st = ov.stats()
        res.free()
        st["wall_ms"] = dt * 1e3
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
