#!/usr/bin/env python3
"""Developer probe: what a PCIe copy running on another stream does to the two big kernels.

The resident call (reads in HBM, kernels only) at config 2, its counting pass and verify kernel timed by the library's own
HIP events -- alone, beside a device->host copy loop, beside a host->device copy loop (page-locked torch tensors, a torch
stream of their own, copies of --mb megabytes back to back for the whole call).  The streamed step runs exactly this
mixture: piece k's records go home while piece k + 1 is scanned.

    python tools/copy_beside_kernel.py --mb 8 --iters 6
"""
import argparse
import json
import os
import statistics
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from phasm_amd import synth  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--mb", type=int, default=8)
    ap.add_argument("--iters", type=int, default=6)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ov = ExactOverlapper(device=0)
    for name, seq in synth.oriented(synth.generate_reads(synth.CONFIGS[a.config])):
        ov.add_sequence(name, seq)
    ov.upload()
    os.environ["PHASM_PHASE_EVENTS"] = "1"
    n = a.mb << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    side = torch.cuda.Stream()

    def call():
        res = ov.overlaps_result(1000)
        st = ov.stats()
        res.free()
        return st

    for _ in range(2):
        call()
    out = {}
    for mode in ("alone", "beside device->host copies", "beside host->device copies"):
        scan, ver, tot, copied = [], [], [], []
        for _ in range(a.iters):
            stop = threading.Event()
            count = [0]

            def pump():
                torch.cuda.set_device(0)
                with torch.cuda.stream(side):
                    while not stop.is_set():
                        for _ in range(4):   # (a few copies queued ahead, so that the link never waits for the host)
                            if mode.startswith("beside device"):
                                host.copy_(dev, non_blocking=True)
                            else:
                                dev.copy_(host, non_blocking=True)
                            count[0] += 1
                        side.synchronize()

            th = None
            if mode != "alone":
                th = threading.Thread(target=pump)
                th.start()
                while count[0] < 8:
                    pass
            c0 = count[0]
            st = call()
            c1 = count[0]
            if th is not None:
                stop.set()
                th.join()
            scan.append(st["ms_scan_probe"])
            ver.append(st["ms_verify_kernel"] if st.get("ms_verify_kernel") else st["ms_verify"])
            tot.append(st["ms_total"])
            copied.append((c1 - c0) * a.mb)
        out[mode] = {"ms_scan_probe": round(statistics.median(scan), 3), "ms_verify": round(statistics.median(ver), 3),
                     "ms_total": round(statistics.median(tot), 3), "MB_copied_during_the_call": statistics.median(copied)}
        print(mode, json.dumps(out[mode]), flush=True)
    base = out["alone"]
    for mode, v in out.items():
        if mode != "alone":
            print("%s: counting pass x %.2f, verify x %.2f" % (mode, v["ms_scan_probe"] / base["ms_scan_probe"], v["ms_verify"] / base["ms_verify"]))
    ov.close()


if __name__ == "__main__":
    main()
