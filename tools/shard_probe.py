#!/usr/bin/env python3
"""Developer probe: what one rank's share of a step costs at N = 1, 2, 4, 8 (one GPU runs shard k of N):
wall time and device time of po_candidates_shard, then po_expand of the concatenated candidates."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--min-length", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--index-only", action="store_true",
                    help="time the whole index build against one rank's sub-table of the sliced index (large read sets)")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    ov = ExactOverlapper()
    for name, seq in synth.oriented(synth.generate_reads(cfg)):
        ov.add_sequence(name, seq)
    ov.upload()
    if a.index_only:
        os.environ["PHASM_NO_INDEX_REUSE"] = "1"
        r = ov.overlaps_result(a.min_length)
        r.free()
        r = ov.overlaps_result(a.min_length)
        st = ov.stats()
        r.free()
        print(json.dumps({"whole_index_ms": round(st["ms_index"], 3), "wide": st["wide_index"], "step_ms": round(st["ms_total"], 3)}))
        for ns in (2, 4, 8):
            t, ex = [], []
            for it in range(a.iters):
                t0 = time.perf_counter()
                wide, bits, entries = ov.index_slice_build(a.min_length, ns - 1, ns)
                t1 = time.perf_counter()
                chunk = ov.index_chunk_bytes(bits, entries)
                buf = torch.empty(chunk, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                ov.index_slice_export(buf.data_ptr(), entries)
                t3 = time.perf_counter()
                t.append((t1 - t0) * 1e3)
                ex.append((t3 - t2) * 1e3)
                del buf
            print(json.dumps({"n_slices": ns, "slice_build_wall_ms": round(min(t), 3), "slice_index_device_ms": round(ov.stats()["ms_index"], 3),
                              "export_ms": round(min(ex), 3), "chunk_MB": round(chunk / 1e6, 1),
                              "all_gather_MB_in_per_rank": round(chunk * (ns - 1) / 1e6, 1)}))
        ov.close()
        sys.exit(0)
    for ns in (1, 2, 4, 8):
        for k in sorted({0, ns // 2, ns - 1}):
            wall, dev, n = [], [], 0
            for it in range(a.iters + 1):
                t0 = time.perf_counter()
                r = ov.candidates_result(a.min_length, k, ns)
                dt = time.perf_counter() - t0
                st = ov.stats()
                n = len(r)
                r.free()
                if it:
                    wall.append(dt * 1e3)
                    dev.append(st["ms_total"])
            print(json.dumps({"nshards": ns, "shard": k, "cands": n, "wall_ms": round(float(np.mean(wall)), 3),
                              "device_ms": round(float(np.mean(dev)), 3),
                              "stages": {x: round(st[x], 3) for x in ("ms_index", "ms_scan_count", "ms_scan_fill", "ms_verify", "ms_select", "ms_emit")}}))
    # expansion of the full candidate set (what every rank does after the all-gather)
    parts = []
    for k in range(8):
        r = ov.candidates_result(a.min_length, k, 8)
        t = torch.empty((len(r), 4), dtype=torch.int32, device="cuda")
        r.copy_to_device(t.data_ptr())
        r.free()
        parts.append(t)
    allc = torch.cat(parts)
    torch.cuda.synchronize()
    for it in range(a.iters):
        t0 = time.perf_counter()
        r = ov.expand_result(allc.data_ptr(), allc.shape[0])
        dt = time.perf_counter() - t0
        n = len(r)
        r.free()
    print(json.dumps({"expand_rows": n, "expand_wall_ms": round(dt * 1e3, 3), "expand_device_ms": round(ov.stats()["ms_emit"], 3)}))
    ov.close()
