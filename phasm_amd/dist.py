"""Multi-GPU form of the overlap step: a-side shards + one all-gather.

The reference is one process on one core (no collective exists in it).  The path shards
naturally on the *a-side*: every rank holds the whole packed read set and anchor table
(24 G bases at 2 bit = 6 GB for the largest BASELINE config, trivial against 288 GB of HBM3E)
and scans a contiguous range of reads balanced by base count.  All row rules are local to ``a``
(longest-only per (a,b), every containment occurrence), so the only exchange is merging the
per-rank results: one RCCL all-gather over xGMI (``torch.distributed`` backend "nccl" is RCCL on
ROCm).  Rank order = read order: the merged result is the same array for every rank count and the same
multiset of rows as the single-GPU result (sharded calls pick the canonical member of a strand-mirror pair
by a scrambled read order, which balances the shards' verify work).

What travels is the compact form, not the rows: each rank's *verified candidates*
(``po_candidates_shard``: 16 bytes each and, in paired-strand mode, one per strand-mirror pair --
46 MB instead of 168 MB of rows at BASELINE config 2), and every rank expands the merged array
into rows locally (``po_expand``, ~0.1 ms).  xGMI is point-to-point, so an all-gather is bound
by the per-link rate: shrinking the payload is what scales.

RCCL has no all-gatherv: counts are gathered first, then one padded ``all_gather_into_tensor``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from ._lib import CAND_DTYPE, ROW_DTYPE


def merge_row_shards(local: torch.Tensor, group=None, keep_padding: bool = False) -> torch.Tensor:
    """All-gather variable-length ``int32[n_local, k]`` tensors (k = 6 rows / 4 candidates); every
    rank gets the concatenation in rank order.  Any backend (RCCL on GPU, gloo on CPU).

    ``keep_padding``: return the gathered buffer as it is -- ``world_size`` equal slots of ``max_n`` entries, each
    shard followed by all-zero entries -- instead of squeezing the padding out (one copy kernel per rank).
    ``po_expand`` skips all-zero candidates, so the candidate exchange uses this form."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    ws = dist.get_world_size(group)
    dev = local.device
    k = local.shape[1]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(ws, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts_h = counts.tolist()
    max_n = max(counts_h)
    if max_n == 0:
        return local
    padded = local
    if local.shape[0] != max_n:
        padded = torch.zeros((max_n, k), dtype=torch.int32, device=dev)
        padded[: local.shape[0]] = local
    gathered = torch.empty((ws * max_n, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    if keep_padding or all(c == max_n for c in counts_h):
        return gathered
    return torch.cat([gathered[r * max_n: r * max_n + c] for r, c in enumerate(counts_h)], dim=0)


def _result_to_tensor(res, k: int, device: torch.device) -> torch.Tensor:
    n = len(res)
    out = torch.empty((n, k), dtype=torch.int32, device=device)
    if n:
        if device.type == "cuda":
            res.copy_to_device(out.data_ptr())  # device-to-device, no host round trip
        else:
            out.copy_(torch.from_numpy(res.rows().view(np.int32).reshape(-1, k)))
    return out


def local_shard_rows(ov, min_length: int, rank: int, world_size: int, device: torch.device) -> torch.Tensor:
    """This rank's rows as ``int32[n, 6]`` on ``device``."""
    res = ov.overlaps_result(min_length, rank, world_size)
    try:
        return _result_to_tensor(res, 6, device)
    finally:
        res.free()


def local_shard_candidates(ov, min_length: int, rank: int, world_size: int, device: torch.device) -> torch.Tensor:
    """This rank's verified candidates as ``int32[n, 4]`` (a, p, b, type) on ``device``."""
    res = ov.candidates_result(min_length, rank, world_size)
    try:
        return _result_to_tensor(res, 4, device)
    finally:
        res.free()


def expand_candidates(ov, cands: torch.Tensor):
    """Rows (an ``OverlapResult``, device resident) from a merged candidate tensor on the GPU."""
    if cands.device.type != "cuda":
        cands = cands.cuda()
    cands = cands.contiguous()
    # the library works on its own HIP stream: the collective / concatenation that produced `cands`
    # must have finished on torch's stream before it reads them
    torch.cuda.current_stream(cands.device).synchronize()
    res = ov.expand_result(cands.data_ptr(), cands.shape[0])
    return res


def sharded_overlaps(ov, min_length: int, group=None, device: Optional[torch.device] = None) -> torch.Tensor:
    """Every rank returns the full merged ``int32[n_rows, 6]`` row tensor (on its GPU)."""
    if dist.is_available() and dist.is_initialized():
        rank, ws = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, ws = 0, 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    merged = merge_row_shards(local_shard_candidates(ov, min_length, rank, ws, device), group, keep_padding=True)
    res = expand_candidates(ov, merged)
    try:
        return _result_to_tensor(res, 6, device)
    finally:
        res.free()


def rows_tensor_to_struct(t: torch.Tensor) -> np.ndarray:
    """int32[n,6] tensor -> structured row array (a_idx, b_idx, astart, aend, bstart, bend)."""
    return np.ascontiguousarray(t.cpu().numpy()).view(ROW_DTYPE).reshape(-1)


def cands_tensor_to_struct(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.cpu().numpy()).view(CAND_DTYPE).reshape(-1)
