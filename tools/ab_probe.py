"""Developer probe: A/B of environment settings on the host-to-host step, INTERLEAVED -- box-to-box and minute-to-minute
noise (+-3 %) is larger than most single changes to the streamed step, so settings take turns inside one process and the
medians over all rounds are compared.

    python tools/ab_probe.py "" "PHASM_STREAM_CUTS=160,315,455,585,705,810,895,963" "PHASM_HOME_THREADS=6" --rounds 6

A setting is a space-separated list of NAME=VALUE (empty string = defaults).  PHASM_HOME_THREADS and other settings read
once per process only count in the process that starts with them: use one invocation per value for those."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--with-torch", action="store_true", help="import torch and initialise its CUDA context first (what bench.py's process looks like)")
    args = ap.parse_args()
    cfg = synth.CONFIGS[args.config]
    if args.with_torch:
        import torch
        torch.cuda.set_device(0)
        torch.cuda.synchronize()
        print("torch threads", torch.get_num_threads(), flush=True)
    ov = ExactOverlapper(device=0)
    for name, seq in synth.oriented(synth.generate_reads(cfg)):
        ov.add_sequence(name, seq)

    def step():
        t0 = time.perf_counter()
        ov.invalidate()
        res = ov.overlaps_to_host_result(1000)
        n = len(res.rows_view())
        res.free()
        return (time.perf_counter() - t0) * 1e3, n

    times = {s: [] for s in args.settings}
    touched = set()
    for rnd in range(args.rounds):
        for s in args.settings:
            for k in touched:
                os.environ.pop(k, None)
            for kv in s.split():
                k, v = kv.split("=", 1)
                os.environ[k] = v
                touched.add(k)
            for _ in range(3):     # (the first steps after a change of cuts re-learn the candidate counts and buffers)
                step()
            for _ in range(args.steps):
                times[s].append(step()[0])
    for s in args.settings:
        t = sorted(times[s])
        print("%-70s median %.3f  min %.3f  p25 %.3f  p75 %.3f  p95 %.3f  max %.3f  mean %.3f  (%d steps)"
              % (s or "(defaults)", statistics.median(t), t[0], t[len(t) // 4], t[3 * len(t) // 4], t[min(len(t) - 1, len(t) * 95 // 100)], t[-1],
                 sum(t) / len(t), len(t)), flush=True)
    ov.close()


if __name__ == "__main__":
    main()
