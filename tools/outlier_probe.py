#!/usr/bin/env python3
"""Developer probe: catch a slow host-to-host step and show where its time went (PHASM_HOME_TRACE per piece: when the records
were home, when the rows were written), next to a normal step.

    python tools/outlier_probe.py --steps 400 --slow 6.0
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--slow", type=float, default=6.0)
    a = ap.parse_args()
    os.environ["PHASM_HOME_TRACE"] = "1"
    ov = ExactOverlapper(device=0)
    for name, seq in synth.oriented(synth.generate_reads(synth.CONFIGS["cfg2"])):
        ov.add_sequence(name, seq)
    tmp = tempfile.TemporaryFile()
    saved = os.dup(2)
    shown_normal = False
    n_slow = 0
    for it in range(a.steps):
        tmp.seek(0)
        tmp.truncate()
        os.dup2(tmp.fileno(), 2)
        t0 = time.perf_counter()
        ov.invalidate()
        res = ov.overlaps_to_host_result(1000)
        n = len(res.rows_view())
        res.free()
        ms = (time.perf_counter() - t0) * 1e3
        os.dup2(saved, 2)
        if it < 5:
            continue
        if ms > a.slow or not shown_normal:
            tmp.seek(0)
            print("---- step %d: %.3f ms%s" % (it, ms, " (SLOW)" if ms > a.slow else " (a normal one)"), flush=True)
            print(tmp.read().decode("utf-8", "replace"), flush=True)
            shown_normal = True
            n_slow += ms > a.slow
    print("%d slow steps of %d" % (n_slow, a.steps - 5))
    ov.close()


if __name__ == "__main__":
    main()
