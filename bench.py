#!/usr/bin/env python3
"""Benchmark of the overlap hot path on MI355X: overlaps/s (+ read-pairs/s) on BASELINE.json's
config 2 -- 50k x 15 kb error-free reads from a 5 Mb diploid genome, both strands added as
`phasm overlap` does (100k oriented reads), min-overlap 1000.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = ONE call of the hot path the way the reference's `overlaps()` is one call: it
starts from the packed reads in HOST memory and ends with the 24-byte row array in HOST memory --
`po_invalidate` + `po_upload` (H2D of the packed read set) + `po_overlaps` (N=1) or `po_candidates_shard` + RCCL
all-gather + `po_expand` (N>1; fixed total work = strong scaling) + `po_result_rows` (D2H of the rows).  `value`
is rows / that wall time.  The same step with the reads already resident in HBM and the rows left there is
reported as `resident` (the kernel pipeline alone, never `value`); building the reference API's list of Python
tuples from the row array is reported as `python_tuples`.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (k_verify): algorithmic bytes of the emitted overlaps
                (sum 2*ceil(l/4) B, both sides at 2 bit/base) / its HIP-event duration, vs 8 TB/s
  cpu_baseline  the reference overlapper itself (oracle/_ref, built from /root/reference by
                `make -C oracle ref`) timed on this host's cores on a bounded, same-density sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from phasm_amd import synth  # noqa: E402
from phasm_amd.dist import CandidateExchange, ReadExchange, expand_candidates  # noqa: E402
from phasm_amd.overlapper import ExactOverlapper  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(min_length: int, sample_reads: int) -> dict:
    """Time the CPU baseline on a bounded sample with cfg2's coverage per haplotype (so the same
    overlaps per read).  This is the only place bench.py touches oracle/ -- as the thing compared
    against, never as part of the measured GPU path."""
    from oracle import overlap_oracle as oo
    cfg = synth.scaled(synth.CONFIGS["cfg2"], sample_reads)
    seqs = [s for _, s in synth.oriented(synth.generate_reads(cfg))]
    sample = ("cfg2 density at %d reads: %d oriented x %d b, G=%d diploid snp %.3f seed %d, min_length %d"
              % (cfg.n_reads, len(seqs), cfg.read_len, cfg.genome_len, cfg.snp, cfg.seed, min_length))
    if oo.have_reference():
        _, secs, nrows = oo.reference_overlaps(seqs, min_length, quiet=True)
        kind = "reference"
    else:
        arr, secs = oo.oracle_overlaps_struct(seqs, min_length)
        nrows = len(arr)
        kind = "port"
    n = len(seqs)
    return {"value": nrows / secs if secs > 0 else None, "unit": "overlaps/s", "cores": 1, "kind": kind,
            "sample": sample, "seconds": round(secs, 3), "rows": int(nrows),
            "read_pairs_per_sec": n * (n - 1) / secs if secs > 0 else None,
            "host_cores_available": os.cpu_count()}


def measured_hbm_gbs(device) -> float:
    """What this box's HBM actually delivers: a device-to-device copy of 2 GiB (read + write counted)."""
    n = 1 << 31
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del src, dst
    return 2.0 * n / (ms * 1e-3) / 1e9


def measured_peaks() -> dict:
    """The hardware rates the kernels are priced against, measured now (tools/ubench.hip through tools/ubench.py)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ubench
        lib = ubench.load()
        valu = max(lib.ub_valu(op, 8, 40000) for op in (1, 2))      # v_xor_b32 / v_lshrrev_b32, 8 waves per SIMD
        return {"valu_wave_insts_per_sec_chip": valu * 1024,
                "valu_cycles_per_inst_at_2.4GHz": 2.4e9 / valu,
                "valu_vop3_wave_insts_per_sec_chip": lib.ub_valu(0, 8, 40000) * 1024,   # v_alignbit_b32: VOP3 forms issue at half rate
                "lds_random_b64_TBps": lib.ub_lds(8, 4000) / 1e12,
                "l2_hit_TBps": lib.ub_l2(8, 2000, 2 << 20) / 1e12,
                "fabric_TBps": lib.ub_l2(8, 1000, 64 << 20) / 1e12,
                # random 64-byte lines per second beyond an XCD's L2 (a table probe that misses L2: one line per lane), from the
                # Infinity Cache (64 MB table) and from HBM (2 GB table); and from L2 itself (4 MB table)
                "random_lines_G_per_s": {"4_MB": lib.ub_rand_lines(4 << 20, 8, 300, 1, 8) / 1e9,
                                         "64_MB": lib.ub_rand_lines(64 << 20, 8, 300, 1, 8) / 1e9,
                                         "2048_MB": lib.ub_rand_lines(2048 << 20, 8, 300, 1, 8) / 1e9}}
    except Exception as e:  # noqa: BLE001 -- the bench line must come out even if the microbenchmarks cannot run
        return {"error": repr(e)}


def cfg4_leg(dev_idx: int, m: int, max_diff: int, band: int) -> dict:
    """BASELINE config 4 (config 2 + 1 % substitution noise) both ways: the exact path (parity with the reference:
    it finds nothing there) and the banded seed-extension DP (po_overlaps_ex; an extension beyond the reference,
    parity unpinned -- oracle/extend_oracle.c is its checker)."""
    cfg = synth.CONFIGS["cfg4"]
    ov = ExactOverlapper(device=dev_idx)
    for name, seq in synth.oriented(synth.generate_reads(cfg)):
        ov.add_sequence(name, seq)
    ov.upload()
    out = {"workload": "cfg4: %d x %d b reads, %.0f %% substitution noise, both strands, min_overlap %d"
                       % (cfg.n_reads, cfg.read_len, cfg.noise * 100, m)}
    for key, call in (("exact", lambda: ov.overlaps_result(m)),
                      ("banded_dp", lambda: ov.overlaps_ex_result(m, max_diff, band))):
        call().free()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = call()
        n = len(res)
        dt = time.perf_counter() - t0
        res.free()
        st = ov.stats()
        out[key] = {"ms_per_call": dt * 1e3, "rows": n, "candidates": st["n_candidates"], "overlaps_per_sec": n / dt,
                    "ms_verify_kernel": st["ms_verify_kernel"]}
        if key == "banded_dp":
            cells = st["dp_steps"] * (2 * band + 1) / 2.0     # half of the band's lanes hold a cell of each antidiagonal
            out[key].update({"max_diff": max_diff, "band": band, "antidiagonals": st["dp_steps"], "stopped_early": st["dp_stopped"],
                             "G_cell_updates_per_sec": cells / (st["ms_verify_kernel"] * 1e-3) / 1e9,
                             "kernel": {2: "k_extend_bits: one LANE per candidate, the band row as a bit vector (Myers / Hyyro diagonal band: 20 vector instructions per row "
                                           "whatever the band, rows in branch-free blocks of 16), bases fetched 64 at a time one round ahead, candidates sorted by length",
                                        1: "k_extend_lanes: one LANE per candidate (band row in registers, 64 candidates per wave, candidates sorted by length)",
                                        0: "k_extend_dp<2>: one WAVE per candidate, one lane per diagonal, antidiagonal sweep with whole-wave DPP shifts"}[st["dp_lanes"]],
                             "parity": "unpinned (the reference is exact); checker: oracle/extend_oracle.c"})
    ov.close()
    return out


def layout_leg(ov: ExactOverlapper, m: int, with_cpu: bool) -> dict:
    """Next row of the path: stage 1 of `phasm layout` -- classify, contained-read
    and alignment filters, assembly-graph edges -- on the rows of one step, still resident in HBM."""
    res = ov.overlaps_result(m)
    for _ in range(2):
        e, _r = ov.layout_edges(res, want_removed=False)
        e.free()
    K = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = 0.0
    for _ in range(K):
        e, _r = ov.layout_edges(res, want_removed=False)
        acc += ov.layout_stats()["ms_total"]
        e.free()
    dt = (time.perf_counter() - t0) / K
    st = ov.layout_stats()
    algo = 24 * st["n_rows"] + 16 * st["n_edges"]           # every row read once, every edge written once
    traffic = None
    try:   # HBM-side bytes of the layout kernels from the rocprofv3 --pmc passes (profiles/traffic.json)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            lt = [v for k, v in json.load(f).get("cfg2", {}).items() if k.startswith("k_layout_") or k.startswith("k_count_bytes")]
        traffic = int(sum(lt)) if lt else None
    except (OSError, ValueError):
        pass
    out = {"rows_per_sec": st["n_rows"] / dt, "ms_per_call": dt * 1e3, "device_ms": acc / K,
           "n_rows": st["n_rows"], "n_edges": st["n_edges"], "n_contained_reads": st["n_contained_reads"],
           "stage_ms": {k: round(st[k], 4) for k in ("ms_classify", "ms_dedupe", "ms_emit")},
           "roofline": {"bound": "hbm", "kernel": "k_layout_classify + k_layout_winner_adjacent + k_layout_emit (streaming: rows in, edges out; no dedupe table for rows of the paired-strand emission)",
                        "achieved": algo / (acc / K * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo / (acc / K * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_GBps": traffic / (acc / K * 1e-3) / 1e9 if traffic else None,
                        "algorithmic_bytes_per_call": int(algo)}}
    if with_cpu:
        from oracle import layout_oracle as lo
        from oracle import overlap_oracle as oo
        rows = oo.struct_to_rows(res.rows()[:1_000_000])
        L = ov.lengths()
        t1 = time.perf_counter()
        lo.layout_vectorised(rows, L)
        tv = time.perf_counter() - t1
        t1 = time.perf_counter()
        lo.layout_sequential(rows[:100_000], L)
        ts = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": len(rows) / tv, "unit": "rows/s", "cores": 1, "kind": "port",
                               "sample": "first %d rows of the same step, numpy restatement (oracle/layout_oracle.py)" % len(rows),
                               "literal_python_rows_per_sec": 100_000 / ts}
    res.free()
    return out


def cli_overlap_leg(reads, m: int) -> dict:
    """`python -m phasm_amd.cli overlap reads.fasta -l m -o out.gfa` in a fresh process: interpreter start, GPU runtime
    start, FASTA ingest, the ONE cold overlap call, GFA2 written -- the wall time of the whole command."""
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="phasm_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fa, out = os.path.join(d, "reads.fasta"), os.path.join(d, "overlaps.gfa")
        synth.write_fasta(fa, reads)
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, "-m", "phasm_amd.cli", "overlap", "--timing", fa, "-l", str(m), "-o", out],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, text=True)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr[-500:]}
        stages = None
        for ln in p.stderr.splitlines():
            if ln.startswith("PHASM_CLI_TIMING "):
                stages = json.loads(ln[len("PHASM_CLI_TIMING "):])
        if stages is not None:
            stages["process exit + launch overhead (wall - the stages above)"] = round(dt - sum(stages.values()), 4)
        n_e = 0
        with open(out, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                n_e += chunk.count(b"\nE\t")
        return {"cli_overlap_seconds": dt, "stages_seconds": stages, "fasta_bytes": os.path.getsize(fa), "gfa_bytes": os.path.getsize(out), "e_lines": n_e,
                "note": "wall time of the child process `python -m phasm_amd.cli overlap` (files in /dev/shm): interpreter + GPU runtime start, "
                        "FASTA ingest, one cold po_overlaps call, S and E lines written"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_throttled():
    """(nr_throttled, throttled_usec) of this process's cgroup (v2 cpu.stat), or None."""
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def other_config_leg(dev_idx: int, name: str, m: int, steps: int = 3) -> dict:
    """Another BASELINE config on ONE GPU, outside the headline's timed region: the host-to-host step (a changed
    read set: streamed upload, kernels, rows home -- what `value` measures at config 2) and the kernels alone (reads resident,
    rows left in HBM).  BASELINE.json names configs 3 and 5 for 8 GPUs; this is their single-GPU number."""
    cfg = synth.CONFIGS[name]
    t0 = time.time()
    ov = ExactOverlapper(device=dev_idx)
    for rname, seq in synth.oriented(synth.generate_reads(cfg)):
        ov.add_sequence(rname, seq)
    t_load = time.time() - t0
    h2h, rows, st = [], 0, {}
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ov.invalidate()
        res = ov.overlaps_to_host_result(m)
        rows = len(res.rows_view())
        dt = (time.perf_counter() - t1) * 1e3
        res.free()
        if it:
            h2h.append(dt)
            st = ov.stats()
    resident = []
    res_st = {}
    for it in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = ov.overlaps_result(m)
        n2 = len(res)
        dt = (time.perf_counter() - t1) * 1e3
        res.free()
        if it:
            resident.append(dt)
            res_st = ov.stats()
    assert n2 == rows
    ov.close()
    keys = ("ms_index", "ms_scan_count", "ms_scan_fill", "ms_verify", "ms_select", "ms_emit", "ms_total", "ms_scan_probe", "ms_verify_kernel")
    return {"workload": "%s: %d x %d b error-free reads, %d b %d-ploid genome, both strands, min_overlap %d, ONE GPU"
                        % (name, cfg.n_reads, cfg.read_len, cfg.genome_len, cfg.ploidy, m),
            "rows": int(rows), "host_to_host_ms": min(h2h), "host_to_host_ms_all": [round(x, 3) for x in h2h],
            "overlaps_per_sec": rows / (min(h2h) * 1e-3), "streamed": int(st.get("streamed", 0)), "wide_index": int(st.get("wide_index", 0)),
            "upload_bytes": int(st.get("upload_bytes", 0)), "upload_ms": st.get("ms_upload"),
            "resident_ms": min(resident), "resident_overlaps_per_sec": rows / (min(resident) * 1e-3),
            "resident_stage_ms": {k: round(res_st.get(k, 0.0), 4) for k in keys},
            "index_reused_in_resident_calls": int(res_st.get("index_reused", 0)),
            "load_seconds": round(t_load, 1)}


def cold_call_leg(dev_idx: int, oriented, m: int, n_handles: int = 3) -> dict:
    """The first call on a FRESH handle -- what the reference's one-shot caller sees (the index is built inside the one call,
    overlapper.cpp:33-36; `phasm overlap` makes exactly one, assembler.py:42): po_create ... reads added (untimed, like
    addSequence) ... then po_overlaps_to_host + po_result_rows timed, next to the second call on the same handle."""
    cold, second = [], []
    rows = 0
    for _ in range(n_handles):
        ov = ExactOverlapper(device=dev_idx)
        for name, seq in oriented:
            ov.add_sequence(name, seq)
        torch.cuda.synchronize()
        for dst in (cold, second):
            if dst is second:
                ov.invalidate()
            t0 = time.perf_counter()
            res = ov.overlaps_to_host_result(m)
            rows = len(res.rows_view())
            dst.append((time.perf_counter() - t0) * 1e3)
            res.free()
        ov.close()
    return {"cold_call_ms": min(cold), "cold_call_ms_all": [round(x, 3) for x in cold], "second_call_ms": [round(x, 3) for x in second],
            "rows": rows, "note": "fresh handle each time (this process has run the step before): po_create, reads added untimed, "
                                  "then ONE po_overlaps_to_host + po_result_rows timed"}


def visible_gpu_count() -> int:
    """GPUs this process may use, counted WITHOUT the HIP runtime: the KFD topology lists every node (GPUs are the nodes
    with SIMDs), ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow it down.  The launcher parent
    must stay a process that has never initialised the GPU (it starts the ranks as children)."""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return 0 if not os.path.isdir("/sys/class/kfd") else -1   # (no KFD driver: no GPU; nodes unreadable: unknown)
    for f in files:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) as CHILD processes and pass rank 0's
    JSON line on.  The parent never initialises the GPU (a process that has must not exec or fork GPU work), it only counts
    devices; with --dist-backend gloo the ranks share whatever GPUs there are (a rehearsal, merged on the host)."""
    import socket
    import subprocess
    n_dev = visible_gpu_count()      # (from the kernel driver's topology files: no HIP / torch call in this process)
    if args.dist_backend == "nccl" and 0 <= n_dev < args.gpus:
        print("bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one device per rank; --dist-backend gloo rehearses "
              "the N-rank step on fewer)" % (args.gpus, n_dev), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's peer mappings need it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print("bench.py: the %d-rank run failed (exit code %d)" % (args.gpus, p.returncode), file=sys.stderr)
        return p.returncode or 1
    print(line, flush=True)
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--reads", type=int, default=0, help="scale the config to this many reads (dev only)")
    ap.add_argument("--min-length", type=int, default=1000)
    ap.add_argument("--cpu-sample-reads", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tuples", action="store_true", help="skip the Python tuple materialisation leg")
    ap.add_argument("--no-stream", action="store_true", help="N=1: upload the whole read set first (po_upload), then call po_overlaps_to_host (the unstreamed form of the step)")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the config-4 leg (exact path + banded DP)")
    ap.add_argument("--other-configs", default="cfg3,cfg5", help="comma-separated BASELINE configs run after the headline legs on one GPU "
                                                           "(host-to-host step and kernels alone; '' = none; cfg5 adds ~35 s and tens of GB of host memory)")
    ap.add_argument("--no-cli", action="store_true", help="skip the `overlap` command leg (a child process: FASTA file in, GFA2 file out)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-call leg (first call on fresh handles)")
    ap.add_argument("--dist-path", action="store_true",
                    help="dev: run the N>1 code path (shard + RCCL all-gather + expansion) even with one rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 path with several ranks on one GPU (rows merged on the host)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process never touches the GPU, it starts the N ranks as a child
        # (torch.distributed.run) and relays rank 0's JSON line
        return launch_ranks(args)

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner on stdout when its first
    # communicator comes up, and libraries may print what they like -- everything written to descriptor 1 from
    # here on goes to stderr, the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    cfg = synth.CONFIGS[args.config]
    if args.reads:
        cfg = synth.scaled(cfg, args.reads)
    reads_once = synth.generate_reads(cfg)
    # the `overlap` COMMAND as a user runs it -- one process, one call (assembler.py:42): FASTA file in, GFA2 file out.
    # Run as a child BEFORE this process touches the GPU (a process that has initialised the GPU starts no children).
    cli_leg = None
    if world == 1 and not args.dist_path and not args.no_cli:
        cli_leg = cli_overlap_leg(reads_once, args.min_length)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the overlap path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_idx = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_idx)
    device = torch.device("cuda", dev_idx)
    merge_device = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1 or args.dist_path:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        # how many ranks really take part in a collective (goes into the line as `rccl_ranks`)
        ones = torch.ones(1, dtype=torch.int64, device=merge_device)
        dist.all_reduce(ones)
        ranks_in_collective = int(ones.item())
        assert ranks_in_collective == dist.get_world_size() == world
    else:
        ranks_in_collective = 1

    t_load = time.time()
    oriented_once = synth.oriented(reads_once)
    del reads_once
    ov = ExactOverlapper(device=dev_idx)
    for name, seq in oriented_once:
        ov.add_sequence(name, seq)
    ov.upload()  # packed reads resident in HBM before anything is timed
    n_oriented = len(ov)
    t_load = time.time() - t_load
    m = args.min_length

    exchange = CandidateExchange(ov, device=merge_device) if (world > 1 or args.dist_path) else None
    read_exchange = ReadExchange(ov, device=device) if world > 1 else None
    stage_keys = ["ms_index", "ms_scan_count", "ms_scan_fill", "ms_verify", "ms_select", "ms_emit", "ms_total", "ms_scan_probe", "ms_verify_kernel"]
    acc = {k: 0.0 for k in stage_keys}
    last = {}

    pcie = {"h2d_s": 0.0, "d2h_s": 0.0}

    def step(timed: bool, inclusive: bool = True) -> int:
        if inclusive:
            # the call starts from host memory: packed reads host -> device (po_invalidate + po_upload)
            t_a = time.perf_counter()
            ov.invalidate()
            if read_exchange is not None:
                read_exchange.upload()    # N > 1: 1/N of the packed reads over this rank's PCIe link, the rest over xGMI
            elif args.no_stream or args.dist_path:
                ov.upload()               # (the unstreamed form: the whole read set first, then the kernels)
            # else: po_overlaps_to_host finds the read set changed and streams it up itself, piece by piece under the kernels
            if timed:
                pcie["h2d_s"] += time.perf_counter() - t_a
        if world == 1 and not args.dist_path:
            # (host to host: the pipelined call -- chunk k's rows travel while chunk k + 1 is computed)
            res = ov.overlaps_to_host_result(m) if inclusive else ov.overlaps_result(m)
        elif inclusive:
            # the pipelined N-rank step (phasm_amd/dist.py, CandidateExchange.rows_home): this rank's candidates go into the
            # all-gather (the merged list ends up on every GPU), and WHILE they travel the rank expands its own candidates
            # into rows and brings them home -- rank order = read order, the ranks' arrays together are the merged rows
            merged, res = exchange.rows_home(m)
            shard_st = ov.stats()
            del merged
        else:
            # exchange the compact form (verified candidates, 16 B), expand to rows on every rank
            merged = exchange.candidates(m)   # shard + one all-gather of fixed slots (phasm_amd/dist.py)
            shard_st = ov.stats()          # stage timings of this rank's shard (before the expansion)
            res = expand_candidates(ov, merged)
        n = len(res)
        if inclusive:
            # ... and ends in host memory: the row array device -> host (po_result_rows; a view, no second copy)
            t_a = time.perf_counter()
            rows = res.rows_view()     # (N > 1: this rank's own rows; already home, rows_home copied them under the collective)
            assert len(rows) == n
            del rows
            if timed:
                pcie["d2h_s"] += time.perf_counter() - t_a
        res.free()
        st = ov.stats()
        if world > 1 or args.dist_path:
            if not inclusive:
                # the expansion saw every rank's candidates: scale its byte counters to this rank's share
                for k in ("verify_bytes_algo", "sum_overlap_bases", "n_rows"):
                    st[k] = st[k] // world
            for k in stage_keys:
                if k not in ("ms_emit",):
                    st[k] = shard_st[k]
            st["n_candidates"] = shard_st["n_candidates"]
            st["shard_bases"] = shard_st["shard_bases"]
        if timed:
            for k in stage_keys:
                acc[k] += st[k]
            if inclusive and st.get("streamed"):
                pcie["h2d_s"] += st["ms_upload"] * 1e-3   # first piece's copy starts -> last piece has landed (device events)
                pcie["streamed_steps"] = pcie.get("streamed_steps", 0) + 1
        last.update(st)
        return n

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # (loading the reads used every core the process may have: on a box with a CPU quota the scheduler may still owe the
    # process a throttled period -- cpu.stat nr_throttled counts them --, and a period that starts inside the timed loop
    # stops every thread of the step for milliseconds.  Let it pass before the clock starts.)
    time.sleep(0.3)
    # (the process holds a few hundred thousand Python objects -- the generated reads --: a generation-2 pass of the cyclic
    # collector inside a 4.7 ms step is a pause of milliseconds that has nothing to do with the step)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step(False)
    fence()
    throttled0 = cpu_throttled()
    t0 = time.perf_counter()
    n_rows = 0
    step_ms = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        n_rows = step(True)
        step_ms.append((time.perf_counter() - t_s) * 1e3)   # (a step ends with its rows in host memory: nothing is in flight here)
    fence()
    dt = time.perf_counter() - t0
    throttled1 = cpu_throttled()
    gc.enable()
    if world > 1 or args.dist_path:
        t = torch.tensor([dt], dtype=torch.float64, device=merge_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank brought ITS rows home: the job's rows per step are the sum (outside the timed region)
        t = torch.tensor([n_rows], dtype=torch.int64, device=merge_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_rows = int(t.item())
    incl_acc = dict(acc)
    incl_last = dict(last)   # (statistics of the last host-to-host step, before the resident loop overwrites them)
    # The pieces of a streamed step record no timing events inside the timed region (a marker between two kernels costs ~5 us
    # of device time, seven per piece): the per-kernel sums of the SAME step are taken from three extra steps with
    # PHASM_PHASE_EVENTS=2 (events around the two big kernels and around each piece), outside the timed region.
    ev_steps = 3
    if world == 1 and not args.dist_path and incl_last.get("streamed"):
        keep_pcie = dict(pcie)
        for k in acc:
            acc[k] = 0.0
        os.environ["PHASM_PHASE_EVENTS"] = "2"
        for _ in range(ev_steps):
            step(True)
        os.environ.pop("PHASM_PHASE_EVENTS")
        fence()
        for k in ("ms_scan_probe", "ms_verify_kernel", "ms_total"):
            incl_acc[k] = acc[k] * args.steps / ev_steps     # (scaled to the K steps the averages below divide by)
        pcie.clear()
        pcie.update(keep_pcie)
        step(False)   # (one step without the events again: the statistics of `last` are those of the timed form)
    # the kernel pipeline alone (reads resident in HBM, rows left in HBM): an extra, never `value`
    for k in acc:
        acc[k] = 0.0
    step(False, inclusive=False)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(True, inclusive=False)
    fence()
    dt_res = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt_res], dtype=torch.float64, device=merge_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_res = float(t.item())

    out = None
    if rank == 0:
        K = max(args.steps, 1)
        ms_step = dt / K * 1e3
        avg = {k: incl_acc[k] / K for k in stage_keys}      # HIP events of the kernels INSIDE the timed region
        avg_res = {k: acc[k] / K for k in stage_keys}       # ... and of the resident loop
        ver_bytes = last["verify_bytes_algo"]                      # this rank's shard, per launch
        ver_gbs = ver_bytes / (avg["ms_verify"] * 1e-3) / 1e9 if avg["ms_verify"] > 0 else 0.0
        job_bytes = last["shard_bases"] * last["bits_per_base"] / 8 + ver_bytes + 24 * last["n_rows"]
        job_gbs = job_bytes / (avg["ms_total"] * 1e-3) / 1e9 if avg["ms_total"] > 0 else 0.0
        out = {
            "metric": "overlaps_per_sec", "value": n_rows / (dt / K), "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            # (this rank's steps one by one: ms_per_step is the mean the contract asks for; a box whose host cores are busy
            # with someone else's work shows as a max far from the median -- the step's last stage runs on host threads)
            "steps_ms": {"median": sorted(step_ms)[len(step_ms) // 2], "min": min(step_ms), "max": max(step_ms),
                         "p90": sorted(step_ms)[min(len(step_ms) - 1, (len(step_ms) * 9) // 10)],
                         "all": [round(x, 3) for x in step_ms],
                         # (CPU-quota periods in which the scheduler stopped this process, and for how long, inside the timed loop)
                         "cpu_quota_throttled_periods": None if throttled0 is None or throttled1 is None else throttled1[0] - throttled0[0],
                         "cpu_quota_throttled_ms": None if throttled0 is None or throttled1 is None else (throttled1[1] - throttled0[1]) / 1e3},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d x %d b error-free reads, %d b %d-ploid genome (snp %.3f, seed %d), "
                                   "both strands = %d oriented reads, min_overlap %d"
                                   % (args.config, cfg.n_reads, cfg.read_len, cfg.genome_len, cfg.ploidy,
                                      cfg.snp, cfg.seed, n_oriented, m),
                       "n_reads": cfg.n_reads, "read_len": cfg.read_len, "min_length": m,
                       "timed_region": (("host to host per step: po_invalidate + po_overlaps_to_host (the changed read set is streamed host -> device piece by piece while "
                                         "the pieces that have arrived go through the kernels and their rows travel device -> host) + po_result_rows"
                                         if not args.no_stream else
                                         "host to host per step: po_invalidate + po_upload (H2D of the packed reads) + po_overlaps_to_host (kernels, D2H of the rows pipelined chunk by chunk) + po_result_rows")
                                        if world == 1 and not args.dist_path else
                                        "host to host per step, every rank: po_invalidate + sharded upload in parts (po_upload_piece_part: 1/N of the packed reads over this rank's PCIe "
                                        "link, the all-gather of part k over xGMI under the copy of part k + 1, po_upload_assemble_parts) + po_candidates_shard_into + ONE all-gather of "
                                        "verified candidates (the merged list on every GPU) and, while it is in flight, po_expand of this rank's own candidates + po_result_rows "
                                        "(this rank's rows to its host; the ranks' arrays together are the merged rows)"),
                       "parallelism": "each rank uploads 1/N of the packed reads in parts, RCCL all-gathers over xGMI (async, under the next part's PCIe copy) complete every rank's copy; a-side read shards x%d + one RCCL all-gather of verified candidates (16 B, fixed slots) per step, under which each rank expands and brings home the rows of its own shard (1/%d of the job's rows)" % (world, world) if world > 1
                                      else "single GPU"},
            "rows_per_step": int(n_rows),
            "rccl_ranks": ranks_in_collective if args.dist_backend == "nccl" else 0,
            "dist_backend": (args.dist_backend if (world > 1 or args.dist_path) else None),
            "ranks": ranks_in_collective,
            "read_pairs_per_sec": n_oriented * (n_oriented - 1) / (dt / K),
            "resident": {"overlaps_per_sec": n_rows / (dt_res / K), "ms_per_step": dt_res / K * 1e3,
                         "stage_ms": {k: round(v, 4) for k, v in avg_res.items()},
                         "index_reused": int(last.get("index_reused", 0)),
                         "note": "same step with the packed reads already in HBM and the rows left in HBM; index_reused = 1: the anchor index of the "
                                 "unchanged upload is kept across these calls (ms_index is then the reset only; PHASM_NO_INDEX_REUSE=1 rebuilds it per call: "
                                 "0.06 ms narrow, 2.9 / 16 ms for the wide index of configs 3 / 5)"},
            "candidates_per_step": int(last["n_candidates"]),
            "stage_ms": {k: round(v, 4) for k, v in avg.items()},
            "load_seconds": round(t_load, 1),
        }
        # ---- roofline.  The contract's figure -- SURVEY.md section 8d's algorithmic bytes / launch time -- is kept as
        # `algorithmic_GBps`, but it is not a fraction of anything for these kernels: the verify reads a from LDS and b
        # from L2, compares a strand-mirror pair once and emits it twice.  Each kernel is priced instead against the
        # four things it can saturate, with peaks MEASURED on this box (tools/ubench.hip, run live below):
        #   hbm   bytes on the memory side of L2 (PMC FETCH_SIZE / WRITE_SIZE, profiles/traffic.json) vs 8 TB/s
        #   l2    bytes the kernel loads from L2 vs the measured L2-hit rate of global_load_dwordx4
        #   lds   bytes it reads from LDS vs the measured rate of random ds_read_b64
        #   valu  VALU wave-instructions (PMC SQ_INSTS_VALU) vs the measured issue rate of 2-operand integer ops
        # `bound` = the largest fraction; every fraction is <= 1 by construction.
        bits = last["bits_per_base"]
        scan_bytes = last["shard_bases"] * bits // 8
        exec_bytes = last["verify_bytes_exec"]          # both sides of every VERIFIED CANDIDATE, once
        sharded = "true" if world > 1 else "false"
        # (kernel names as rocprofv3 prints them; the trailing `false` = not the streamed step's instantiation: the resident
        # loop and the --pmc passes run whole-set launches)
        scan_name = ("k_wide_scan<%d, false>" % bits) if last["wide_index"] else ("k_scan_probe<%d, true, false>" % bits)
        ver_name = "k_verify_a<%d, %s, true, false>" % (bits, sharded)
        peaks = measured_peaks() if world == 1 else {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f) if not args.reads and world == 1 else {}
        except (OSError, ValueError):
            tj = {}
        pmc_traffic = tj.get(args.config, {})
        pmc_ctr = tj.get(args.config + "_counters", {})
        try:   # the VALU model: cycles per instruction class (calibrated), static mix per kernel, padding sensitivity
            with open(os.path.join(ROOT, "profiles", "r03_valu_model.json")) as f:
                valu_model = json.load(f)["kernels"] if args.config == "cfg2" and world == 1 else {}
        except (OSError, ValueError, KeyError):
            valu_model = {}
        positions = last["shard_bases"]                  # one filter lookup (8 B of LDS) per position of the scan
        # (timed_region_sum_ms: summed over the pieces of a host-to-host step of the same form, taken with PHASM_PHASE_EVENTS=2
        # in three extra steps right behind the timed loop -- the timed steps themselves record no per-kernel events)
        # Launch durations: HIP events around the kernel in the RESIDENT loop (one whole-set launch per step, the same
        # launches the rocprofv3 --pmc passes profile); inside the host-to-host region the same work runs as 4 chunk
        # launches, whose summed duration is reported next to it (`timed_region_sum_ms`).
        kern = {
            scan_name: {"avg_launch_ms": avg_res["ms_scan_probe"], "timed_region_sum_ms": avg["ms_scan_probe"],
                        "algorithmic_bytes_per_launch": int(scan_bytes),
                        "l2_bytes": None if last["wide_index"] else int(scan_bytes),   # (+ the table groups, counted on the hbm side)
                        "lds_bytes": None if last["wide_index"] else int(positions * 8)},
            ver_name: {"avg_launch_ms": avg_res["ms_verify_kernel"], "timed_region_sum_ms": avg["ms_verify_kernel"],
                       "algorithmic_bytes_per_launch": int(ver_bytes),
                       "executed_compare_bytes": int(exec_bytes),
                       "l2_bytes": int(exec_bytes // 2),             # the b side streams from L2 (locality order)
                       "lds_bytes": int(exec_bytes // 2 * 5 // 4)},  # the a side: 5 dwords read per 4 compared
        }
        for name, v in kern.items():
            t = v["avg_launch_ms"] * 1e-3
            v["algorithmic_GBps"] = v["algorithmic_bytes_per_launch"] / t / 1e9 if t > 0 else 0.0
            v["traffic"] = pmc_traffic.get(name)
            lim = {}
            if v["traffic"] and t > 0:
                lim["hbm"] = {"GBps": v["traffic"] / t / 1e9, "peak_GBps": HBM_PEAK_GBS}
            sp = tj.get(args.config + "_split", {}).get(name)
            rl = peaks.get("random_lines_G_per_s") if isinstance(peaks, dict) else None
            if sp and rl and t > 0:
                # the scan's table probes that miss L2 are random 64-byte line requests (calibrated split of FETCH_SIZE,
                # tools/pmc_to_traffic.py): priced against the rate this memory system gives such requests, measured above
                # (the narrow table's misses are served by the Infinity Cache, the wide index's by HBM)
                pk = rl["2048_MB"] if last["wide_index"] else rl["64_MB"]
                lim["fabric_lines"] = {"G_lines_per_s": sp["random_64B_lines"] / t / 1e9, "peak_G_lines_per_s": pk,
                                       "random_64B_lines_per_launch": sp["random_64B_lines"], "stream_bytes": sp["stream_bytes"]}
                v["traffic_uncalibrated_2x_rule"] = sp.get("uncalibrated_2x_rule_bytes")
            if v.get("l2_bytes") and peaks.get("l2_hit_TBps") and t > 0:
                lim["l2"] = {"GBps": v["l2_bytes"] / t / 1e9, "peak_GBps": peaks["l2_hit_TBps"] * 1e3}
            if v.get("lds_bytes") and peaks.get("lds_random_b64_TBps") and t > 0:
                lim["lds"] = {"GBps": v["lds_bytes"] / t / 1e9, "peak_GBps": peaks["lds_random_b64_TBps"] * 1e3}
            ctr = pmc_ctr.get(name, {})
            vm = valu_model.get(name, {})
            if ctr.get("SQ_INSTS_VALU") and ctr.get("SQ_BUSY_CU_CYCLES") and vm.get("static_mean_cycles_per_valu_inst"):
                # VALU busy = instructions x their issue cost / SIMD-cycles of the launch.  The cost per instruction is the
                # kernel's static mix priced with the cycles tools/valu_calib.py measured (2.15 for VOP1/VOP2 forms, 4.05 for
                # VOP3 integer forms -- SQ_ACTIVE_INST_VALU turned out to count instructions, not cycles, so it cannot say).
                # In units of G SIMD-cycles per second: the peak is every SIMD busy every cycle of the launch.
                simd_cycles = 4.0 * ctr["SQ_BUSY_CU_CYCLES"]
                busy_cycles = ctr["SQ_INSTS_VALU"] * vm["static_mean_cycles_per_valu_inst"]
                lim["valu"] = {"G_simd_cycles_per_s": busy_cycles / t / 1e9, "peak_G_simd_cycles_per_s": simd_cycles / t / 1e9,
                               "wave_insts_per_launch": ctr["SQ_INSTS_VALU"],
                               "mean_cycles_per_inst_static_mix": vm["static_mean_cycles_per_valu_inst"],
                               "sensitivity": vm.get("valu_sensitivity"),
                               "sensitivity_note": "tools/pad_probe.sh: kernel cycles gained per full-rate VALU cycle added to the hot loop "
                                                   "(1 = VALU issue is what the kernel waits for, 0 = the pipe had room)"}
            for k2, x in lim.items():
                ach = x.get("GBps", x.get("G_simd_cycles_per_s", x.get("G_lines_per_s")))
                pk = x.get("peak_GBps", x.get("peak_G_simd_cycles_per_s", x.get("peak_G_lines_per_s")))
                x["frac"] = ach / pk
            v["limits"] = lim
            if lim:
                v["bound"] = max(lim, key=lambda k2: lim[k2]["frac"])
                v["frac"] = lim[v["bound"]]["frac"]
        dom = max(kern, key=lambda k: kern[k]["avg_launch_ms"])
        d = kern[dom]
        bl = d["limits"].get(d.get("bound", ""), {})
        out["roofline"] = {"bound": d.get("bound"), "kernel": dom,
                           "achieved": bl.get("GBps", bl.get("G_simd_cycles_per_s", bl.get("G_lines_per_s"))),
                           "peak": bl.get("peak_GBps", bl.get("peak_G_simd_cycles_per_s", bl.get("peak_G_lines_per_s"))),
                           "unit": ("G SIMD-cycles/s (VALU busy)" if d.get("bound") == "valu" else
                                    "G random 64-B lines/s" if d.get("bound") == "fabric_lines" else "GB/s"),
                           "valu_sensitivity": bl.get("sensitivity"),
                           "frac": d.get("frac"), "traffic": d["traffic"],
                           "hbm_frac": d["limits"].get("hbm", {}).get("frac"),
                           "avg_launch_ms": d["avg_launch_ms"],
                           "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                           "algorithmic_GBps": d["algorithmic_GBps"],
                           "algorithmic_note": "SURVEY 8d bytes / launch time; above the HBM peak because a comes from LDS, b from L2, and a "
                                               "strand-mirror pair is compared once and emitted twice -- not a roofline fraction",
                           "measured_peaks": peaks,
                           "pmc_source": "profiles/traffic.json (rocprofv3 --pmc passes, tools/collect_pmc.sh + tools/pmc_to_traffic.py) and "
                                         "profiles/r03_valu_model.json (tools/valu_calib.py, tools/isa_mix.py, tools/pad_probe.sh): `traffic`, the VALU "
                                         "instruction and CU-cycle counts and the VALU model are replayed from there, everything else is measured in this run",
                           "kernels": kern,
                           "job_algorithmic_bytes": int(job_bytes), "job_algorithmic_GBps": job_gbs}
        if world == 1:
            h2d_bytes = float(last["upload_bytes"])   # (half the packed set when the odd reads are rebuilt on the device)
            d2h_bytes = 24.0 * n_rows
            out["pcie"] = {"h2d_bytes_per_step": int(h2d_bytes), "packed_read_set_bytes": int(last["total_bases"] * last["bits_per_base"] / 8),
                           "h2d_ms": pcie["h2d_s"] / K * 1e3,
                           "h2d_GBps": h2d_bytes / (pcie["h2d_s"] / K) / 1e9 if pcie["h2d_s"] > 0 else None,
                           "d2h_bytes_per_step": int(d2h_bytes),
                           # what really crossed device -> host: one record per strand-mirror pair (8 or 16 bytes), the rows are
                           # written by the library's host threads (po_stats.home_record_bytes; 0 = the rows themselves crossed)
                           "d2h_record_bytes": int(incl_last.get("home_record_bytes", 0)),
                           "d2h_wire_bytes_per_step": int(incl_last.get("home_record_bytes", 0) * incl_last.get("n_verified", 0)
                                                          if incl_last.get("home_record_bytes", 0) else d2h_bytes),
                           "link_h2d_GBps_measured": 57.0,   # (tools/h2d_source_probe.hip: one 188 MB copy; 54.5 in 11 pieces with events)
                           "kernels_plus_d2h_ms": (dt - pcie["h2d_s"]) / K * 1e3,
                           "d2h_alone_ms_at_measured_h2d_rate": d2h_bytes / (h2d_bytes / (pcie["h2d_s"] / K)) * 1e3 if pcie["h2d_s"] > 0 else None,
                           "peak_GBps_per_direction": 64.0,
                           "pcie_floor_ms": (h2d_bytes + d2h_bytes) / 64e9 * 1e3,
                           "pcie_floor_duplex_ms": max(h2d_bytes, d2h_bytes) / 64e9 * 1e3,
                           "upload_floor_ms_at_measured_link_rate": h2d_bytes / 57.0e9 * 1e3,
                           "streamed": bool(pcie.get("streamed_steps")),
                           "deferred_containments": int(incl_last.get("n_deferred", 0)),
                           "predicted_pieces": int(incl_last.get("n_predicted", 0)), "fused_tails": int(incl_last.get("fused_tail", 0)),
                           "note": ("streamed step: the packed reads go up piece by piece (h2d_ms = first copy starts -> last piece landed, device events) "
                                    "while the pieces that have arrived are scanned, verified and emitted and their rows travel home -- PCIe carries both "
                                    "directions at once, so kernels_plus_d2h_ms overlaps h2d_ms and the two do not add up to ms_per_step. "
                                    if pcie.get("streamed_steps") else
                                    "h2d = wall time of po_upload inside the timed region; the D2H of the rows is pipelined behind the "
                                    "kernels inside po_overlaps_to_host (chunk k travels while chunk k + 1 is computed); ") +
                                   "PCIe Gen5 x16 = 64 GB/s per direction; pcie_floor = both transfers at that rate back to back, pcie_floor_duplex = the longer of the two alone"}
            if pcie.get("streamed_steps"):
                out["pcie"]["kernels_plus_d2h_ms"] = None
                out["pcie"]["after_upload_ms"] = dt / K * 1e3 - pcie["h2d_s"] / K * 1e3   # what is left of a step once the last piece has landed
            if not args.no_tuples:
                # what the reference API returns: a list of (id_a, id_b, astart, aend, bstart, bend) tuples
                t1 = time.perf_counter()
                tup = ov.overlaps(m)
                dt_t = time.perf_counter() - t1
                out["python_tuples"] = {"seconds": dt_t, "tuples": len(tup), "tuples_per_sec": len(tup) / dt_t,
                                        "note": "ExactOverlapper.overlaps(): one step + building the list of Python tuples"}
                del tup
            out["roofline"]["hbm_copy_measured"] = measured_hbm_gbs(device)   # GB/s of a d2d copy on this box
            out["layout_stage1"] = layout_leg(ov, m, not args.no_cpu_baseline)
            if not args.no_cold:
                cc = cold_call_leg(dev_idx, oriented_once, m)
                out["cold_call_ms"] = cc["cold_call_ms"]
                out["cold_call"] = cc
                out["cold_over_steady"] = cc["cold_call_ms"] / ms_step
            if cli_leg is not None:
                out["cli_overlap_seconds"] = cli_leg.get("cli_overlap_seconds")
                out["cli_overlap"] = cli_leg
            if not args.no_cfg4 and not args.reads:
                out["cfg4_extension"] = cfg4_leg(dev_idx, m, 400, 8)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(m, args.cpu_sample_reads)
            if args.other_configs and not args.reads:
                # (last: the headline's handle is closed first so that the big configs have the GPU's memory to themselves)
                ov.close()
                del oriented_once
                out["other_configs"] = {}
                for name in [x.strip() for x in args.other_configs.split(",") if x.strip()]:
                    try:
                        out["other_configs"][name] = other_config_leg(dev_idx, name, m)
                    except Exception as e:  # noqa: BLE001 -- the headline line must come out
                        out["other_configs"][name] = {"error": repr(e)}
    if world > 1 or args.dist_path:
        dist.barrier()
        dist.destroy_process_group()
    ov.close()
    sys.stdout.flush()
    if out is not None:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    return 0


if __name__ == "__main__":
    sys.exit(main())
