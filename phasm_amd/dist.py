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

RCCL has no all-gatherv: ``merge_row_shards`` gathers counts first, then one padded ``all_gather_into_tensor``;
``CandidateExchange`` is the steady-state form with fixed slots and a single collective per step.
"""
from __future__ import annotations

import os
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


class ReadExchange:
    """Upload of the packed reads for an N-rank job: 1/N over each rank's PCIe link, the rest over xGMI.

    Every rank needs the whole read set in HBM, and uploading all of it on every rank (190 MB at config 2 at 52 GB/s =
    3.6 ms) is the largest term of a host-to-host step that does not shrink with N.  Here rank g copies the words of
    shard g's even reads into its slot (``po_upload_piece``), ONE all-gather of equal slots moves the pieces (RCCL has
    no all-gatherv: slots are the longest piece, agreed without talking -- every rank knows every piece's length from
    its own copy of the read table), and ``po_upload_assemble`` puts them in place and rebuilds the odd reads on the
    device.  Falls back to the plain upload for read sets that are not (x, reverse complement of x) pairs."""

    def __init__(self, ov, group=None, device: Optional[torch.device] = None):
        self.ov = ov
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.ws = dist.get_world_size(group) if self.on else 1
        self.device = device if device is not None and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self.via_host = self.on and dist.get_backend(group) != "nccl"
        self.local = None
        self.gathered = None
        self.n_collectives = 0
        self.n_parts = 1
        self._plan = None      # ((reads, parts asked for), parts, slot words) of the current read set

    def upload(self, parts: Optional[int] = None) -> bool:
        """True: sharded upload done.  False: plain ``po_upload`` was used.

        The piece travels in ``parts`` parts (default: one per ~8 MB of this rank's piece, at most 4): the host->device copy
        of part k + 1 runs while the all-gather of part k is in flight (``async_op``), so the step costs about
        ``max(PCIe, xGMI)`` plus one part instead of their sum."""
        if not self.on or self.ws < 2:
            self.ov.upload()
            return False
        key = (len(self.ov), parts)
        if self._plan is None or self._plan[0] != key:
            # (once per state of the read set: every rank computes every piece's length from its own read table)
            sizes = [self.ov.upload_piece(k, self.ws) for k in range(self.ws)]
            if not all(ok for ok, _ in sizes):
                self._plan = (key, None, 0)
            else:
                longest = max(n for _, n in sizes)
                np_ = parts
                if np_ is None:
                    np_ = int(os.environ.get("PHASM_UPLOAD_PARTS", "0")) or max(1, min(4, (longest * 8) >> 23))
                np_ = max(1, min(int(np_), 64))
                # equal slots: the longest part of any shard
                slot = max(self.ov.upload_piece_part(k, self.ws, p, np_)[1] for k in range(self.ws) for p in range(np_)) + 1
                self._plan = (key, np_, slot)
        _, parts, slot = self._plan
        if parts is None:
            self.ov.upload()
            return False
        if self.local is None or self.local.shape != (parts, slot):
            self.local = torch.empty((parts, slot), dtype=torch.int64, device=self.device)
            self.gathered = torch.empty((parts, self.ws * slot), dtype=torch.int64, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
        pending = []
        for p in range(parts):
            # host -> device, this rank's link (returns when the part has landed; the all-gather of the part before is in flight)
            self.ov.upload_piece_part(self.rank, self.ws, p, parts, self.local[p].data_ptr(), slot)
            if self.via_host:
                host = torch.empty(self.ws * slot, dtype=torch.int64)
                dist.all_gather_into_tensor(host, self.local[p].cpu(), group=self.group)
                self.gathered[p].copy_(host)
            else:
                pending.append(dist.all_gather_into_tensor(self.gathered[p], self.local[p], group=self.group, async_op=True))   # xGMI
            self.n_collectives += 1
        for w in pending:
            w.wait()
        torch.cuda.current_stream(self.device).synchronize()
        self.n_parts = parts
        self.ov.upload_assemble(self.gathered.data_ptr(), slot, self.ws, parts)
        return True


class IndexExchange:
    """The anchor index of large read sets, built once per node instead of once per rank.

    The wide index (more than 160 k reads of length >= min_length) is GBs of table at BASELINE config 5 and used to
    be built whole by every rank -- the replicated part of a sharded step (16 of 62 ms there).  Here rank g builds
    sub-table g of ``world_size`` (keys partitioned by hash, ``po_index_slice_build``), copies it with its chain
    segment into one chunk (``po_index_slice_export``) and ONE all-gather over RCCL/xGMI hands every rank all chunks;
    the shard call then probes the gathered index (``po_candidates_shard_indexed``).  Chunk size is agreed with a
    tiny all-gather of the chain-segment lengths (RCCL has no all-gatherv).  ``get`` caches the gathered index for
    one ``(min_length, number of reads)``; call ``invalidate`` after adding reads.  Returns None when the read set
    uses the narrow index (0.06 ms to build: every rank keeps building its own)."""

    def __init__(self, ov, group=None, device: Optional[torch.device] = None):
        self.ov = ov
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.ws = dist.get_world_size(group) if self.on else 1
        # the index lives on the GPU whatever carries the collective (gloo rehearsals move the chunks through the host)
        self.device = device if device is not None and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self.via_host = self.on and dist.get_backend(group) != "nccl"
        self.key = None
        self.index = None     # dict(buf=uint8 tensor, n_slices, bits, cap) or None
        self.n_collectives = 0

    def invalidate(self) -> None:
        self.key, self.index = None, None

    def get(self, min_length: int):
        key = (int(min_length), len(self.ov))
        if key == self.key:
            return self.index
        self.key, self.index = key, None
        if not self.on or self.ws < 2 or not torch.cuda.is_available():
            return None
        wire = torch.device("cpu") if self.via_host else self.device
        wide, bits, entries = self.ov.index_slice_build(min_length, self.rank, self.ws)
        mine = torch.tensor([entries, bits, 1 if wide else 0], dtype=torch.int64, device=wire)
        every = torch.empty(3 * self.ws, dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(every, mine, group=self.group)
        self.n_collectives += 1
        every = every.view(self.ws, 3).cpu()
        if not bool(every[:, 2].all()):
            return None        # (same reads on every rank: all wide or none)
        assert int(every[:, 1].min()) == int(every[:, 1].max()), "ranks disagree on the sub-table size"
        cap = int(every[:, 0].max())
        chunk = self.ov.index_chunk_bytes(bits, cap)
        local = torch.empty(chunk, dtype=torch.uint8, device=self.device)
        buf = torch.empty(self.ws * chunk, dtype=torch.uint8, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()   # (the library writes on its own stream)
        self.ov.index_slice_export(local.data_ptr(), cap)
        if self.via_host:
            host = torch.empty(self.ws * chunk, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
            buf.copy_(host)
        else:
            dist.all_gather_into_tensor(buf, local, group=self.group)
        self.n_collectives += 1
        torch.cuda.current_stream(self.device).synchronize()   # ... and reads the gathered index on its own stream
        self.index = dict(buf=buf, n_slices=self.ws, bits=bits, cap=cap)
        return self.index


class CandidateExchange:
    """The steady-state form of the N>1 step: shard -> ONE all-gather -> expansion.

    ``merge_row_shards`` needs two collectives per step (counts, then padded data) and a host round trip in
    between.  Here every rank sends a fixed-size slot -- entry 0 is a header that carries its candidate count,
    then its candidates, then zeros -- whose size all ranks derive from the previous step's gathered headers
    (so they always agree on it).  If a shard outgrows the slot the headers say so on every rank at once and the
    step is repeated with a bigger slot.  The first step sizes the slot with a count all-gather."""

    def __init__(self, ov, group=None, device: Optional[torch.device] = None, slack: float = 1.1):
        self.ov = ov
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.ws = dist.get_world_size(group) if self.on else 1
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.slack = slack
        self.slot = 0          # entries per rank, header included
        self.local = None      # int32[slot, 4]
        self.gathered = None   # int32[ws * slot, 4]
        self.hdr_host = None
        self.filled = 0
        self.n_collectives = 0  # (tests look at this)
        self.index = IndexExchange(ov, group, self.device) if torch.cuda.is_available() else None   # sliced wide index: built once per node

    def _resize(self, need: int) -> None:
        old, old_filled = self.local, self.filled
        self.slot = max(int(need * self.slack) + 64, self.slot)
        self.local = torch.zeros((self.slot, 4), dtype=torch.int32, device=self.device)
        self.filled = 0        # entries of self.local[1:] the last step wrote
        if old is not None and old_filled:
            # a bigger slot in the middle of a step: what this rank already put into its slot moves along
            self.local[1:1 + old_filled].copy_(old[1:1 + old_filled])
            self.filled = old_filled
        self.gathered = torch.empty((self.ws * self.slot, 4), dtype=torch.int32, device=self.device)
        pin = self.device.type == "cuda"
        self.hdr_host = torch.empty(self.ws, dtype=torch.int32, pin_memory=pin)
        if pin:
            # the zero fill and the move above run on torch's stream; the library writes into the same rows on ITS
            # stream (created non-blocking: no implicit ordering with torch's) right after this returns
            torch.cuda.current_stream(self.device).synchronize()

    def candidates(self, min_length: int) -> torch.Tensor:
        """All ranks' verified candidates of one step as ``int32[ws * slot, 4]`` (padding and neutralised headers
        are all-zero entries, which ``po_expand`` skips)."""
        written = False
        idx = self.index.get(min_length) if (self.on and self.index is not None) else None
        if idx is not None:
            # large read sets: probe the node-wide sliced index instead of building the whole index on this rank
            dst, cap = (self.local.data_ptr() + 16, self.slot - 1) if (self.slot and self.device.type == "cuda") else (0, 0)
            res, written = self.ov.candidates_result_indexed(min_length, self.rank, self.ws, idx["buf"].data_ptr(), idx["n_slices"],
                                                             idx["bits"], idx["cap"], dst, cap)
        elif self.on and self.slot and self.device.type == "cuda":
            # steady state: the compaction kernel writes this rank's candidates straight into its slot
            res, written = self.ov.candidates_result_into(min_length, self.rank, self.ws, self.local.data_ptr() + 16,
                                                          self.slot - 1)
        else:
            res = self.ov.candidates_result(min_length, self.rank, self.ws)
        try:
            if not self.on:
                return _result_to_tensor(res, 4, self.device)

            def fill(dst: torch.Tensor, take: int) -> None:
                if written:
                    return
                if self.device.type == "cuda":
                    # device-to-device on the library's stream (synchronised on return); the torch ops of
                    # _exchange touch other rows of the buffer
                    res.copy_to_device(dst.data_ptr(), take)
                else:
                    dst[:take].copy_(torch.from_numpy(res.rows()[:take].view(np.int32).reshape(-1, 4)))

            return self._exchange(len(res), fill)
        finally:
            res.free()

    def exchange_tensor(self, local: torch.Tensor) -> torch.Tensor:
        """The same exchange for candidates that already sit in a tensor ``int32[n, 4]`` on ``self.device``."""
        if not self.on:
            return local
        return self._exchange(local.shape[0], lambda dst, take: dst[:take].copy_(local[:take]))

    def _exchange(self, n: int, fill) -> torch.Tensor:
        if self.slot == 0:      # first step: agree on a slot size
            n_local = torch.tensor([n], dtype=torch.int64, device=self.device)
            counts = torch.empty(self.ws, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(counts, n_local, group=self.group)
            self.n_collectives += 1
            self._resize(int(counts.max().item()) + 1)
        while True:
            take = min(n, self.slot - 1)
            if take:
                fill(self.local[1:], take)
            if take < self.filled:    # the buffer starts out zero: only what the last step wrote beyond has to go
                self.local[1 + take:1 + self.filled].zero_()
            self.filled = take
            self.local[0, 0] = n      # header: how many this rank HAS (may exceed what fits); the other fields stay 0
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
            self.n_collectives += 1
            slots = self.gathered.view(self.ws, self.slot, 4)
            self.hdr_host.copy_(slots[:, 0, 0], non_blocking=True)
            slots[:, 0, :] = 0        # headers become padding
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            need = int(self.hdr_host.max().item()) + 1
            if need <= self.slot:     # same headers on every rank: same decision everywhere
                return self.gathered
            self._resize(need)        # a shard outgrew the slot: once more with room

    def rows(self, min_length: int):
        """One full step: the merged rows as an ``OverlapResult`` resident on this rank's GPU."""
        return expand_candidates(self.ov, self.candidates(min_length))

    def rows_home(self, min_length: int):
        """The pipelined step: ``(merged candidates on the GPU, this rank's rows as a host array view + their result)``.

        The rows a rank brings home are the rows of ITS OWN candidates (shard g's canonical candidates and their strand
        mirrors): their expansion and their device->host copy need nothing from the other ranks, so they run while the
        all-gather of the candidates is in flight.  Rank order = read order, so the ranks' row arrays concatenated are the
        merged row multiset, and the merged CANDIDATE list -- north_star's "all-gather to merge the per-GPU lists" -- is on
        every GPU for whoever consumes it next (``po_expand``, layout stage 1).  Before: every rank expanded all N shards
        (N x the work) and copied 1/N of the merged rows only after the collective had finished."""
        if not (self.on and self.slot and self.device.type == "cuda"):
            # first step (slot not sized yet) or a host-side rehearsal: the plain sequence, own rows from the merged list
            merged = self.candidates(min_length)
            res = expand_candidates(self.ov, self._own_slice(merged))
            return merged, res
        idx = self.index.get(min_length) if self.index is not None else None
        dst, cap = self.local.data_ptr() + 16, self.slot - 1
        if idx is not None:
            res, written = self.ov.candidates_result_indexed(min_length, self.rank, self.ws, idx["buf"].data_ptr(), idx["n_slices"],
                                                             idx["bits"], idx["cap"], dst, cap)
        else:
            res, written = self.ov.candidates_result_into(min_length, self.rank, self.ws, dst, cap)
        try:
            n = len(res)
            if not written:
                # the shard outgrew its slot: the sized exchange below sorts that out (rare: one step after the data changed)
                def fill(dst_t: torch.Tensor, take: int) -> None:
                    res.copy_to_device(dst_t.data_ptr(), take)
                merged = self._exchange(n, fill)
                return merged, expand_candidates(self.ov, self._own_slice(merged))
            if n < self.filled:
                self.local[1 + n:1 + self.filled].zero_()
            self.filled = n
            self.local[0, 0] = n
            work = dist.all_gather_into_tensor(self.gathered, self.local, group=self.group, async_op=True)
            self.n_collectives += 1
            # ... and while the candidates travel: this rank's rows (the library's stream; reads self.local[1:1+n] only)
            own = self.ov.expand_result(self.local.data_ptr() + 16, n)
            own.rows_view()                                   # device -> host, page-locked, on the library's copy path
            work.wait()
            slots = self.gathered.view(self.ws, self.slot, 4)
            self.hdr_host.copy_(slots[:, 0, 0], non_blocking=True)
            slots[:, 0, :] = 0
            torch.cuda.current_stream(self.device).synchronize()
            need = int(self.hdr_host.max().item()) + 1
            if need > self.slot:                               # another rank outgrew the slot: repeat with room
                self._resize(need)
                own.free()
                return self.rows_home(min_length)
            return self.gathered, own
        finally:
            res.free()

    def _own_slice(self, merged: torch.Tensor) -> torch.Tensor:
        """This rank's candidates inside a merged slot buffer (or the whole tensor when nothing is distributed)."""
        if not self.on or self.slot == 0 or merged.shape[0] != self.ws * self.slot:
            return merged
        return merged[self.rank * self.slot:(self.rank + 1) * self.slot]


def sharded_overlaps(ov, min_length: int, group=None, device: Optional[torch.device] = None) -> torch.Tensor:
    """Every rank returns the full merged ``int32[n_rows, 6]`` row tensor (on its GPU)."""
    if dist.is_available() and dist.is_initialized():
        rank, ws = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, ws = 0, 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    res = CandidateExchange(ov, group, device).rows(min_length)   # (callers that repeat the step keep the exchange)
    try:
        return _result_to_tensor(res, 6, device)
    finally:
        res.free()


def rows_tensor_to_struct(t: torch.Tensor) -> np.ndarray:
    """int32[n,6] tensor -> structured row array (a_idx, b_idx, astart, aend, bstart, bend)."""
    return np.ascontiguousarray(t.cpu().numpy()).view(ROW_DTYPE).reshape(-1)


def cands_tensor_to_struct(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.cpu().numpy()).view(CAND_DTYPE).reshape(-1)
