"""Developer probe: per-dispatch timeline of ONE streamed step from a `rocprofv3 --kernel-trace` CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/stream_probe.py --steps 3 --cuts default
    python3 tools/piece_timeline.py gpurun_out/kt

Prints the last step's dispatches in start order: start (us from the step's first kernel), duration, the gap to the previous
dispatch's end, the kernel -- and per piece (k_revcomp_store opens one) the sum of kernel time and of the gaps."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the steps: k_scatter_first launches once per streamed step
    starts = [i for i, r in enumerate(rows) if r[2].startswith("po::k_scatter_first") or r[2].startswith("k_scatter_first")]
    if len(starts) < 2:
        print("no streamed steps found in", files)
        return
    lo = starts[-1]
    # step ends before the next non-library kernel burst; take everything after lo
    step = rows[lo - 8 if lo >= 8 else 0:]
    t0 = step[0][0]
    prev_end = None
    piece, acc_k, acc_g, n_k = -1, 0.0, 0.0, 0
    out = []
    for s, e, name in step:
        short = name.split("(")[0].replace("po::", "")[:60]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        if short.startswith("k_revcomp_store"):
            if piece >= 0:
                out.append("   == piece %d: %d kernels, kernel time %.1f us, gaps %.1f us" % (piece, n_k, acc_k, acc_g))
            piece += 1
            acc_k = acc_g = 0.0
            n_k = 0
        acc_k += (e - s) / 1e3
        acc_g += max(gap, 0.0)
        n_k += 1
        out.append("%9.1f  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short))
        prev_end = max(prev_end or e, e)
    out.append("   == piece %d: %d kernels, kernel time %.1f us, gaps %.1f us" % (piece, n_k, acc_k, acc_g))
    print("\n".join(out))


if __name__ == "__main__":
    main()
