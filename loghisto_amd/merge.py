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
  * "reduce_scatter": rank r ends with the merged rows of the names it owns, then
                      extracts only those.  The owner blocks are contiguous name ranges
                      of EQUAL PACKED SIZE (cells), not of equal name count: the same
                      rule as k_merge_plan of the C-ABI front-end (lh_snapshot_merge),
                      so both front-ends give a rank the same [first, last).  On a fully
                      connected xGMI hive this moves 1/world of the all-reduce bytes per
                      link.

uint64 counts are reduced through an int64 view: two's-complement addition is the
same bits.  Dirty ranges are merged with MIN/MAX first; then every row travels with its
OWN merged window, packed back to back (plan_windows), so an outlier sample in one name
widens that row only and extract/clear still visit only occupied spans.  The functions take plain tensors so that the CPU tests can drive
them with oracle-built rows; `merge_snapshot` binds them to a live lh_snapshot.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

NKEYS = 65536


def name_blocks(nrows: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) of `rank` when names are cut into blocks of ceil(nrows / world) ids -- for callers that shard the
    name space without a merge plan (logical shards on one device, tools/c4_sim.py).  NOT the reduce-scatter's
    ownership: that follows the merged windows (plan_windows / owned_rows)."""
    per = (nrows + world - 1) // world
    lo = min(rank * per, nrows)
    return lo, min(lo + per, nrows)


def owned_rows(W: dict, rank: int) -> Tuple[int, int]:
    """[first, last) rows owned by `rank` under the plan W = plan_windows(merged ranges, world, plan)."""
    if W["nblocks"] == 1:
        return 0, int(W["width"].shape[0])
    return int(W["brow"][rank]), int(W["brow"][rank + 1])


def merge_ranges(ranges: torch.Tensor, group=None) -> torch.Tensor:
    """ranges: int32[nrows, 2] (lo bin, hi bin); empty rows are (65536, 0). In place."""
    lo = ranges[:, 0].contiguous()
    hi = ranges[:, 1].contiguous()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    ranges[:, 0] = lo
    ranges[:, 1] = hi
    return ranges


def row_bits(rowmax: torch.Tensor, world: int, largest_rank_count: Optional[int] = None) -> torch.Tensor:
    """Bits per cell every row travels at on the C-ABI front-end's wire (k_merge_widths): rowmax[r] = the largest cell any
    rank holds in row r (all-reduced MAX of the per-rank maxima), so world x rowmax[r] bounds every merged cell of the row.
    64 for every row when world x (the largest per-rank sample count of the interval) does not stay below 2^32."""
    if largest_rank_count is not None and not (largest_rank_count < 0xffffffff and largest_rank_count * world < (1 << 32)):
        return torch.full_like(rowmax, 64, dtype=torch.int64)
    bound = rowmax.to(torch.int64).clamp(max=0xffffffff) * world
    bits = torch.full_like(bound, 32)
    bits[bound <= 0xffff] = 16
    bits[bound <= 0xff] = 8
    return bits


def plan_windows(ranges: torch.Tensor, world: int, plan: str, bits: Optional[torch.Tensor] = None):
    """The window plan of a merge, from the MERGED ranges (identical on every rank).

    Every row keeps its own window [lo_r, hi_r]; the windows travel packed back to back, so one outlier
    sample widens one row (at most 65 536 cells), never the whole matrix.  bits[nrows] (row_bits; None: one word per
    cell): the bits per cell a row travels at -- the plan is laid out in WIRE WORDS, row r taking
    ceil(width_r x bits_r / 32) uint32 words (bits 64: width_r words of uint64).  Returns a dict:
      width[nrows]    cells of row r (0 when the row is empty everywhere);  words[nrows] its wire words
      P[nrows + 1]    exclusive prefix of the words (P[nrows] = total words);  Pc[nrows + 1] the same in cells
      nblocks         owner blocks (reduce-scatter: world; all-reduce: 1)
      brow[nb + 1]    first row of every owner block: brow[0] = 0, brow[nb] = nrows, and block k (0 < k < nb) starts
                      at the first row whose prefix reaches k / nb of the total, i.e. the smallest r with
                      P[r] >= total // nb * k + total % nb * k // nb -- contiguous name ranges of equal PACKED size
                      (RCCL's reduce-scatter pads every block to the largest; with equal name counts and names ranked
                      by frequency block 0 holds the widest windows: 1.2 x padding on config 4's slice, 1.0001 x so)
      bstart[nb + 1]  P at the block boundaries;  bmax = largest block, in words;  bstart_c / bmax_c / total_c: in cells
    The SAME rule as k_merge_plan computes on the device for the C-ABI front-end (lh_snapshot_merge returns
    brow[rank], brow[rank + 1]); tests/_stub_merge_driver.py compares the two at 2 .. 8 ranks.
    """
    nrows = ranges.shape[0]
    lo = ranges[:, 0].to(torch.int64)
    hi = ranges[:, 1].to(torch.int64)
    width = (hi - lo + 1).clamp_(min=0)
    if bits is None:
        words = width
    else:
        b = bits.to(torch.int64).to(width.device)
        words = torch.where(b == 64, width, (width * b + 31) >> 5)
    P = torch.zeros(nrows + 1, dtype=torch.int64, device=ranges.device)
    torch.cumsum(words, 0, out=P[1:])
    Pc = torch.zeros(nrows + 1, dtype=torch.int64, device=ranges.device)
    torch.cumsum(width, 0, out=Pc[1:])
    total = int(P[-1].item())
    nb = world if plan == "reduce_scatter" else 1
    k = torch.arange(nb + 1, device=ranges.device, dtype=torch.int64)
    target = total // nb * k + total % nb * k // nb
    brow = torch.searchsorted(P, target, right=False)      # smallest r in [0, nrows] with P[r] >= target
    brow[0], brow[nb] = 0, nrows
    bstart = P[brow]
    bstart_c = Pc[brow]
    bmax = int((bstart[1:] - bstart[:-1]).max().item()) if nb else 0
    bmax_c = int((bstart_c[1:] - bstart_c[:-1]).max().item()) if nb else 0
    return dict(lo=lo, width=width, words=words, P=P, Pc=Pc, nblocks=nb, brow=brow, bstart=bstart, bmax=bmax, total=total,
                bstart_c=bstart_c, bmax_c=bmax_c, total_c=int(Pc[-1].item()))


def block_of_rows(W: dict, rows: torch.Tensor) -> torch.Tensor:
    """Owner block of every row index: the last k with brow[k] <= r (empty blocks share a boundary with their
    successor) -- merge_block_of in lh_kernels.hip."""
    return torch.searchsorted(W["brow"][: W["nblocks"]].contiguous(), rows, right=True) - 1


_buffers = {}


def _buffer(device, n: int) -> torch.Tensor:
    """Grow-only int64 staging buffer per device (no allocation per merge)."""
    key = str(device)
    buf = _buffers.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1) + max(n, 1) // 8, dtype=torch.int64, device=device)
        _buffers[key] = buf
    return buf[:n]


last_info = {}   # what the last merge_rows call on this process moved (tests, bench)


def merge_rows(rows: torch.Tensor, ranges: Optional[torch.Tensor] = None, plan: str = "allreduce",
               group=None, narrow: bool = True) -> Tuple[int, int]:
    """Sum `rows` (int64[nrows, 65536]) across ranks, in place, moving only each row's merged window.

    `rows` may be a strided view: a snapshot's rows are lh_row_stride() cells apart, not 65 536 (snapshot_tensors).
    ranges (int32[nrows, 2], merged in place) bounds the occupied span of every row; None moves whole rows.
    Returns the [first, last) rows that hold fully merged data on this rank."""
    assert rows.dtype == torch.int64 and rows.dim() == 2 and rows.shape[1] == NKEYS and rows.stride(1) == 1
    if plan not in ("allreduce", "reduce_scatter"):
        raise ValueError(f"unknown plan {plan!r}")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nrows = rows.shape[0]
    if ranges is not None:
        merge_ranges(ranges, group)
    else:
        ranges = torch.tensor([[0, NKEYS - 1]], dtype=torch.int32, device=rows.device).repeat(nrows, 1)
    last_info.clear()
    if world == 1:
        return 0, nrows
    # the cells of this rank inside the merged windows, packed position q (0 .. total) -> (row, column)
    lo0 = ranges[:, 0].to(torch.int64)
    width0 = (ranges[:, 1].to(torch.int64) - lo0 + 1).clamp_(min=0)
    total = int(width0.sum().item())
    last_info.update(packed_cells=total, widest_row=int(width0.max().item()) if nrows else 0)
    if total == 0:
        return owned_rows(plan_windows(ranges, world, plan), rank)
    Pc0 = torch.zeros(nrows + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(width0, 0, out=Pc0[1:])
    row_of = torch.repeat_interleave(torch.arange(nrows, device=rows.device), width0, output_size=total)
    q = torch.arange(total, device=rows.device)
    row_stride = rows.stride(0) if nrows > 1 else NKEYS
    flat = row_of * row_stride + (q - Pc0[row_of] + lo0[row_of])
    cells = torch.as_strided(rows, ((nrows - 1) * row_stride + NKEYS,), (1,))   # every cell from row 0 to the last row's end
    mine = cells[flat]
    # the owner blocks are those of the C-ABI front-end: equal shares of the WIRE WORDS, a row travelling at the narrowest
    # of 8 / 16 / 32 bits per cell that holds world x (its largest per-rank cell) -- one more MAX all-reduce here (the
    # C ABI lets it ride with the ranges).  This front-end itself moves int64 cells: only the block boundaries follow.
    # (They follow the C ABI's while its wire words are uint32, i.e. while ranks x the largest per-rank sample count of the
    # interval is below 2^32 -- every stream the tests and the bench merge; beyond that the C ABI sends uint64 words and
    # cuts its blocks on those, which this front-end does not model.)
    bits = None
    if narrow and plan != "allreduce":   # (with one owner block the widths only fed last_info -- no collective for that)
        rowmax = torch.zeros(nrows, dtype=torch.int64, device=rows.device)
        rowmax.scatter_reduce_(0, row_of, mine, "amax", include_self=True)
        dist.all_reduce(rowmax, op=dist.ReduceOp.MAX, group=group)
        bits = row_bits(rowmax, world)
    W = plan_windows(ranges, world, plan, bits)
    own = owned_rows(W, rank)
    last_info.update(packed_words=W["total"])
    if plan == "allreduce":
        buf = _buffer(rows.device, total)
        buf.copy_(mine)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        cells[flat] = buf
        last_info.update(send_bytes=total * 8, recv_bytes=total * 8)
        return own
    # reduce-scatter by contiguous name blocks of equal packed size, every block padded to the largest one
    bmax = W["bmax_c"]
    both = _buffer(rows.device, (world + 1) * bmax)
    send, recv = both[: world * bmax], both[world * bmax:]
    send.zero_()
    blk = block_of_rows(W, row_of)
    send[blk * bmax + (q - W["bstart_c"][blk])] = mine
    dist.reduce_scatter_tensor(recv, send, op=dist.ReduceOp.SUM, group=group)
    sel = blk == rank
    cells[flat[sel]] = recv[(q - W["bstart_c"][blk])[sel]]
    last_info.update(send_bytes=world * bmax * 8, recv_bytes=bmax * 8, padded_words=world * W["bmax"],
                     owned_rows_by_rank=[(int(W["brow"][k]), int(W["brow"][k + 1])) for k in range(world)])
    return own


class _DeviceArray:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 3, "strides": None}


def snapshot_ranges(snap, nrows: int, device: Optional[int] = None):
    """ranges int32[nrows,2] aliasing the snapshot's HBM (the rows are not touched: on an engine of 32-bit cells
    snapshot_tensors moves the snapshot to its uint64 store first, lh_snapshot_rows)."""
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    return torch.as_tensor(_DeviceArray(snap.device_ranges(), (nrows, 2), "<i4"), device=dev)


def snapshot_tensors(snap, nrows: Optional[int] = None, device: Optional[int] = None):
    """(rows int64[nrows,65536] -- a view with row stride lh_row_stride() --, ranges int32[nrows,2]) aliasing the snapshot's HBM."""
    ptr, total = snap.device_rows()
    nrows = total if nrows is None else nrows
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    stride = snap.row_stride()
    flat = torch.as_tensor(_DeviceArray(ptr, ((nrows - 1) * stride + NKEYS,), "<i8"), device=dev)
    rows = torch.as_strided(flat, (nrows, NKEYS), (stride, 1))
    return rows, snapshot_ranges(snap, nrows, device)


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
