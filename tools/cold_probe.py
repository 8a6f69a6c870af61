"""Developer probe: what ONE cold call costs -- the first po_overlaps_to_host + po_result_rows on a fresh handle (what a
`phasm overlap` process does exactly once, /root/reference/phasm/cli/assembler.py:42) -- next to the steady state."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--handles", type=int, default=3)
    ap.add_argument("--trace", action="store_true")
    args = ap.parse_args()
    cfg = synth.CONFIGS[args.config]
    if args.reads:
        cfg = synth.scaled(cfg, args.reads)
    reads = synth.oriented(synth.generate_reads(cfg))
    for hno in range(args.handles):
        ov = ExactOverlapper(device=0)
        t0 = time.perf_counter()
        for name, seq in reads:
            ov.add_sequence(name, seq)
        t_add = time.perf_counter() - t0
        if args.trace and hno in (0, args.handles - 1):
            os.environ["PHASM_ALLOC_TRACE"] = "1"
            os.environ["PHASM_STREAM_TRACE"] = "1"
        times = []
        for call in range(4):
            t0 = time.perf_counter()
            if call:
                ov.invalidate()
            res = ov.overlaps_to_host_result(1000)
            n = len(res.rows_view())
            times.append((time.perf_counter() - t0) * 1e3)
            res.free()
            os.environ.pop("PHASM_ALLOC_TRACE", None)
            os.environ.pop("PHASM_STREAM_TRACE", None)
        print("handle %d: add_sequence %.2f s; calls (ms): %s; rows %d streamed %d" % (
            hno, t_add, " ".join("%.2f" % t for t in times), n, ov.stats()["streamed"]), flush=True)
        ov.close()


if __name__ == "__main__":
    main()
