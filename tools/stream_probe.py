"""Developer probe: host-to-host time of one step (po_invalidate + po_overlaps_to_host + po_result_rows) for several
cuts of the streamed upload, next to the unstreamed form (PHASM_STREAM=0: po_upload, then the chunked call)."""
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
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--cuts", nargs="*", default=["default", "off"])
    ap.add_argument("--trace", action="store_true", help="one more step per setting with PHASM_STREAM_TRACE")
    ap.add_argument("--home-trace", action="store_true", help="one more step per setting with PHASM_HOME_TRACE (no timing events)")
    ap.add_argument("--n-rate", type=float, default=0.0, help="turn this fraction of the bases into N (both strands consistently): exception records")
    args = ap.parse_args()
    cfg = synth.CONFIGS[args.config]
    if args.reads:
        cfg = synth.scaled(cfg, args.reads)
    ov = ExactOverlapper(device=0)
    reads = synth.generate_reads(cfg)
    if args.n_rate > 0:
        import numpy as np
        rng = np.random.default_rng(1)
        out = []
        for name, seq in reads:
            b = bytearray(seq)
            for pos in rng.integers(0, len(b), size=rng.poisson(len(b) * args.n_rate)):
                b[int(pos)] = ord("N")
            out.append((name, bytes(b)))
        reads = out
    for name, seq in synth.oriented(reads):
        ov.add_sequence(name, seq)
    m = cfg.min_overlap if hasattr(cfg, "min_overlap") else 1000
    for cuts in args.cuts:
        os.environ.pop("PHASM_STREAM", None)
        os.environ.pop("PHASM_STREAM_CUTS", None)
        if cuts == "off":
            os.environ["PHASM_STREAM"] = "0"
        elif cuts != "default":
            os.environ["PHASM_STREAM_CUTS"] = cuts

        def step():
            ov.invalidate()
            res = ov.overlaps_to_host_result(m)
            n = len(res.rows_view())
            res.free()
            return n
        for _ in range(3):
            n = step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        st = ov.stats()
        print("cuts %-60s %7.3f ms/step  rows %d  streamed %d  deferred %d  upload %.3f ms  kernels %.3f ms (scan %.3f verify %.3f)  predicted pieces %d  fused tails %d" % (
            cuts, ms, n, st["streamed"], st["n_deferred"], st["ms_upload"], st["ms_total"], st["ms_scan_probe"], st["ms_verify_kernel"],
            st["n_predicted"], st["fused_tail"]), flush=True)
        if args.trace and st["streamed"]:
            os.environ["PHASM_STREAM_TRACE"] = "1"
            step()
            os.environ.pop("PHASM_STREAM_TRACE")
        if args.home_trace and st["streamed"]:
            # when each piece's records were home and its rows written, WITHOUT the timing events a full trace records
            os.environ["PHASM_HOME_TRACE"] = "1"
            t0 = time.perf_counter()
            step()
            print("  (that step: %.3f ms)" % ((time.perf_counter() - t0) * 1e3), flush=True)
            os.environ.pop("PHASM_HOME_TRACE")
    ov.close()


if __name__ == "__main__":
    main()
