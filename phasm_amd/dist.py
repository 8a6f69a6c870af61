"""Multi-GPU form of the overlap step: a-side shards + one all-gather.

The reference is one process on one core (no collective exists in it).  The path shards
naturally on the *a-side*: every rank holds the whole packed read set and anchor table
(24 G bases at 2 bit = 6 GB for the largest BASELINE config, trivial against 288 GB of HBM3E)
and scans a contiguous range of reads balanced by base count (``po_overlaps_shard``).  All row
rules are local to ``a`` (longest-only per (a,b), every containment occurrence), so the only
exchange is merging the per-rank row arrays: one RCCL all-gather over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm).  Rank order = read order, so the merged
array is row-for-row the single-GPU result.

RCCL has no all-gatherv: counts are gathered first, then one padded
``all_gather_into_tensor`` of int32[max_rows, 6]; with 8 ranks on the fully connected xGMI mesh
each peer's shard crosses its own link.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from ._lib import ROW_DTYPE


def merge_row_shards(local_rows: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather variable-length ``int32[n_local, 6]`` row tensors; every rank gets the
    concatenation in rank order.  Works on any backend (RCCL on GPU, gloo on CPU)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_rows
    ws = dist.get_world_size(group)
    dev = local_rows.device
    n_local = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(ws, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts_h = counts.tolist()
    max_n = max(counts_h)
    if max_n == 0:
        return local_rows
    padded = local_rows
    if local_rows.shape[0] != max_n:
        padded = torch.zeros((max_n, 6), dtype=torch.int32, device=dev)
        padded[: local_rows.shape[0]] = local_rows
    gathered = torch.empty((ws * max_n, 6), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    if all(c == max_n for c in counts_h):
        return gathered
    return torch.cat([gathered[r * max_n: r * max_n + c] for r, c in enumerate(counts_h)], dim=0)


def local_shard_rows(ov, min_length: int, rank: int, world_size: int, device: torch.device) -> torch.Tensor:
    """This rank's rows as an ``int32[n, 6]`` tensor on ``device`` (device-to-device copy out of
    the library's result buffer; no host round trip)."""
    res = ov.overlaps_result(min_length, rank, world_size)
    try:
        n = len(res)
        out = torch.empty((n, 6), dtype=torch.int32, device=device)
        if n:
            if device.type == "cuda":
                res.copy_to_device(out.data_ptr())
            else:
                out.copy_(torch.from_numpy(res.rows().view(np.int32).reshape(-1, 6)))
        return out
    finally:
        res.free()


def sharded_overlaps(ov, min_length: int, group=None, device: Optional[torch.device] = None) -> torch.Tensor:
    """Every rank returns the full merged ``int32[n_rows, 6]`` row tensor."""
    if dist.is_available() and dist.is_initialized():
        rank, ws = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, ws = 0, 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return merge_row_shards(local_shard_rows(ov, min_length, rank, ws, device), group)


def rows_tensor_to_struct(t: torch.Tensor) -> np.ndarray:
    """int32[n,6] tensor -> structured row array (a_idx, b_idx, astart, aend, bstart, bend)."""
    return np.ascontiguousarray(t.cpu().numpy()).view(ROW_DTYPE).reshape(-1)
