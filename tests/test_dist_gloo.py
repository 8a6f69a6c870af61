"""N>1 path on CPU: world_size-2 gloo run of the shard + all-gather merge.  Each rank's local
rows are simulated with the CPU oracle restricted to that rank's a-side read range (the same
range po_overlaps_shard scans, taken from the library's own po_shard_range), so this checks the
host-side shard arithmetic and the variable-length gather, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_utils as gu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import overlap_oracle as oo
        from phasm_amd.dist import merge_row_shards, rows_tensor_to_struct
        from phasm_amd.overlapper import ExactOverlapper
        _, seqs, m, want_sorted = gu.ladder_case(case)
        ov = ExactOverlapper()
        for i, s in enumerate(seqs):
            ov.add_sequence("r%d" % i, s)
        lo, hi = ov.shard_range(rank, world)
        whole, _ = oo.oracle_overlaps_struct(seqs, m)            # a-major emission order
        mine = whole[(whole["a_idx"] >= lo) & (whole["a_idx"] < hi)]
        local = torch.from_numpy(mine.view(np.int32).reshape(-1, 6).copy())
        merged = merge_row_shards(local)
        got = rows_tensor_to_struct(merged)
        ok = np.array_equal(got, whole) and np.array_equal(oo.sort_rows(oo.struct_to_rows(got)), want_sorted)
        # the fixed-slot form the candidate exchange uses: every shard followed by all-zero padding
        padded = merge_row_shards(local, keep_padding=True)
        counts = [int(((whole["a_idx"] >= a) & (whole["a_idx"] < b)).sum()) for a, b in (ov.shard_range(r, world) for r in range(world))]
        ok = ok and padded.shape[0] == world * max(counts)
        ok = ok and torch.equal(padded[padded.abs().sum(dim=1) != 0], merged)
        # the one-collective steady-state exchange (CandidateExchange) on the same rows reinterpreted as 4-column
        # entries: step 1 sizes the slot (2 collectives), step 2 reuses it (1), a grown shard forces a resize (2)
        from phasm_amd.dist import CandidateExchange
        ex = CandidateExchange(ov, device=torch.device("cpu"), slack=1.0)
        c4 = torch.from_numpy(np.abs(mine.view(np.int32).reshape(-1, 6)[:, :4]).copy() + 1)   # non-zero entries
        sizes = []
        for rep, mult in enumerate((1, 1, 3)):
            local4 = c4.repeat(mult, 1) if rank == 1 else c4
            before = ex.n_collectives
            g = ex.exchange_tensor(local4)
            sizes.append(ex.n_collectives - before)
            nz = g[g.abs().sum(dim=1) != 0]
            n_mine = [None, None]
            cnt = torch.tensor([local4.shape[0]], dtype=torch.int64)
            allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allc, cnt)
            ok = ok and nz.shape[0] == int(sum(int(x) for x in allc))
            lo_ = 0 if rank == 0 else int(allc[0])
            ok = ok and torch.equal(nz[lo_:lo_ + local4.shape[0]], local4)
        big = len(mine) > 0
        ok = ok and sizes[0] == 2 and sizes[1] == 1 and (sizes[2] == 2 if big else sizes[2] in (1, 2))
        q.put((rank, bool(ok), int(len(mine)), int(len(got))))
        ov.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ladder_small", "ladder_cfg4_noise"])
def test_two_rank_gloo_merge(case):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert sum(n for _, _, n, _ in res) == res[0][3]   # shards partition the rows


def test_merge_is_identity_without_process_group():
    from phasm_amd.dist import merge_row_shards
    t = torch.arange(12, dtype=torch.int32).reshape(2, 6)
    assert merge_row_shards(t) is t
