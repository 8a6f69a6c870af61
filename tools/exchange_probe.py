#!/usr/bin/env python3
"""Developer probe: where the time of one N>1 step goes (one RCCL rank on one GPU, config 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from phasm_amd import synth
from phasm_amd.dist import CandidateExchange, expand_candidates
from phasm_amd.overlapper import ExactOverlapper
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ov = ExactOverlapper(device=0)
for n, s in synth.oriented(synth.generate_reads(synth.CONFIGS["cfg2"])):
    ov.add_sequence(n, s)
ex = CandidateExchange(ov, device=dev)
sync = torch.cuda.synchronize
for it in range(6):
    sync(); t0 = time.perf_counter()
    res = ov.candidates_result(1000, 0, 1); n = len(res)
    sync(); t1 = time.perf_counter()
    g = ex._exchange(n, lambda dst, take: res.copy_to_device(dst.data_ptr(), take))
    sync(); t2 = time.perf_counter()
    res.free()
    rows = expand_candidates(ov, g)
    sync(); t3 = time.perf_counter()
    rows.free()
    sync(); t4 = time.perf_counter()
    print("shard %.3f  exchange %.3f  expand %.3f  free %.3f ms (device shard %.3f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, ov.stats()["ms_emit"]))
dist.destroy_process_group()
