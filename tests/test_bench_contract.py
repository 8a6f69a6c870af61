"""bench.py's contract with the driver: `python bench.py --gpus N` must produce an N-rank line BY ITSELF (the driver's
N=1 command has this shape), and the parent of the ranks must never touch the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multi_rank_launch_fails_loudly_without_devices():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "only 0 GPU(s) visible" in p.stderr and p.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_launches_its_own_ranks(nranks):
    """`python bench.py --gpus N` with no launcher around it: N ranks (gloo rehearsal on this one GPU -- RCCL refuses
    two ranks on one device), one JSON line from rank 0, `ranks` = what an all-reduce of ones counted."""
    import checker as ck
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rc, out, err = ck.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--dist-backend", "gloo",
                           "--reads", "4000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-cfg4", "--no-tuples"],
                          capture_output=True, text=True, env=env, timeout=900)
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["ranks"] == nranks and d["dist_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["metric"] == "overlaps_per_sec" and d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1
    assert d["rows_per_step"] > 100_000


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl_when_two_gpus_are_there():
    """The same self-launch over RCCL proper (one rank per GPU).  Needs two devices: skipped on a one-GPU box -- the
    multi-GPU step has only ever run over gloo on one GPU and as a one-rank RCCL group (README: unmeasured on hardware)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import checker as ck
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rc, out, err = ck.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "4000", "--steps", "2", "--warmup", "1",
                           "--no-cpu-baseline", "--no-cfg4", "--no-tuples"], capture_output=True, text=True, env=env, timeout=900)
    assert rc == 0, err[-3000:]
    d = json.loads([ln for ln in out.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "nccl" and d["rows_per_step"] > 100_000
