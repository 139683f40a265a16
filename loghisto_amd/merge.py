"""Multi-GPU merge of epoch snapshots.

The reference is single-process; this step has no counterpart in it.  Ingest is
data-parallel -- every rank buckets its own slice of the sample stream for all
names, with no per-sample communication -- and the only exchange is the periodic
merge of the uint64 bucket matrix at the epoch flip.  Bucket cells are a
commutative integer sum (metrics.go:278, 292), so the merge is an integer SUM
collective: torch.distributed over RCCL/xGMI on the GPU, gloo on CPU for tests.

Two plans:
  * "allreduce":      every rank ends with every merged row (what BASELINE.json's
                      north star names).
  * "reduce_scatter": rank r ends with the merged rows of the names it owns
                      (contiguous blocks of ceil(M/world) ids), then extracts only
                      those.  On a fully connected xGMI hive this moves 1/world of
                      the all-reduce bytes per link.

uint64 counts are reduced through an int64 view: two's-complement addition is the
same bits.  Dirty ranges are merged with MIN/MAX so extract/clear still visit only
occupied spans.  The functions take plain tensors so that the CPU tests can drive
them with oracle-built rows; `merge_snapshot` binds them to a live lh_snapshot.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

NKEYS = 65536


def owned_rows(nrows: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) rows owned by `rank` under the reduce-scatter plan."""
    per = (nrows + world - 1) // world
    lo = min(rank * per, nrows)
    return lo, min(lo + per, nrows)


def merge_ranges(ranges: torch.Tensor, group=None) -> torch.Tensor:
    """ranges: int32[nrows, 2] (lo bin, hi bin); empty rows are (65536, 0). In place."""
    lo = ranges[:, 0].contiguous()
    hi = ranges[:, 1].contiguous()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    ranges[:, 0] = lo
    ranges[:, 1] = hi
    return ranges


def merge_rows(rows: torch.Tensor, ranges: Optional[torch.Tensor] = None, plan: str = "allreduce",
               group=None, window: Optional[Tuple[int, int]] = None) -> Tuple[int, int]:
    """Sum `rows` (int64[nrows, 65536]) across ranks, in place.

    window=(lo, hi) restricts the exchange to bins [lo, hi] of every row (callers
    obtain it from the merged ranges); None moves whole rows.  Returns the
    [first, last) rows that hold fully merged data on this rank.
    """
    assert rows.dtype == torch.int64 and rows.dim() == 2 and rows.shape[1] == NKEYS
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nrows = rows.shape[0]
    if ranges is not None:
        merge_ranges(ranges, group)
    if world == 1:
        return 0, nrows
    if window is None and ranges is not None:
        lo = int(ranges[:, 0].min().item())
        hi = int(ranges[:, 1].max().item())
        if lo > hi:
            return (0, nrows) if plan == "allreduce" else owned_rows(nrows, rank, world)
        window = (lo, hi)
    full = window is None or (window[0] == 0 and window[1] == NKEYS - 1)
    if plan == "allreduce":
        if full or nrows == 1:
            view = rows if full else rows[:, window[0]: window[1] + 1]
            buf = view if view.is_contiguous() else view.contiguous()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            if buf.data_ptr() != view.data_ptr():
                view.copy_(buf)
        else:
            buf = rows[:, window[0]: window[1] + 1].contiguous()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            rows[:, window[0]: window[1] + 1] = buf
        return 0, nrows
    if plan == "reduce_scatter":
        per = (nrows + world - 1) // world
        w0, w1 = (0, NKEYS - 1) if window is None else window
        width = w1 - w0 + 1
        # pad to world*per rows so every rank contributes equal blocks
        send = torch.zeros((world * per, width), dtype=torch.int64, device=rows.device)
        send[:nrows] = rows[:, w0: w1 + 1]
        recv = torch.empty((per, width), dtype=torch.int64, device=rows.device)
        dist.reduce_scatter_tensor(recv, send, op=dist.ReduceOp.SUM, group=group)
        first, last = owned_rows(nrows, rank, world)
        if last > first:
            rows[first:last, w0: w1 + 1] = recv[: last - first]
        return first, last
    raise ValueError(f"unknown plan {plan!r}")


class _DeviceArray:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 3, "strides": None}


def snapshot_tensors(snap, nrows: Optional[int] = None, device: Optional[int] = None):
    """(rows int64[nrows,65536], ranges int32[nrows,2]) aliasing the snapshot's HBM."""
    ptr, total = snap.device_rows()
    nrows = total if nrows is None else nrows
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    rows = torch.as_tensor(_DeviceArray(ptr, (nrows, NKEYS), "<i8"), device=dev)
    ranges = torch.as_tensor(_DeviceArray(snap.device_ranges(), (nrows, 2), "<i4"), device=dev)
    return rows, ranges


def merge_snapshot(snap, nrows: int, plan: str = "allreduce", group=None) -> Tuple[int, int]:
    """Merge a live snapshot across ranks on torch's current stream, then make the
    snapshot's own stream (extract/clear) wait for it."""
    rows, ranges = snapshot_tensors(snap, nrows)
    cur = torch.cuda.current_stream()
    xs = torch.cuda.ExternalStream(snap.stream())
    cur.wait_stream(xs)            # buffer recycling work queued on the snapshot stream
    first, last = merge_rows(rows, ranges, plan=plan, group=group)
    xs.wait_stream(cur)            # extract must see the merged cells
    return first, last
