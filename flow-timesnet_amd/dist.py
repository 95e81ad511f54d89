"""Batch-sharded multi-GPU TimesBlock forward (one process per GPU, RCCL through
``torch.distributed``).

The reference is single-process; this is new work.  A TimesBlock cannot be sharded
along channels without changing its results (channel median in the selector,
channel-mixing convs), but batch rows are independent once the shared periods are
known.  So every rank owns ``B/world`` rows and the data path needs exactly

1. one tiny exchange per block call: the ``[F]`` fp64 partial batch sums of the
   channel-median spectrum are all-gathered and summed in rank order on every
   rank (deterministic and identical everywhere, unlike a ring all-reduce), which
   makes the selected periods identical on all ranks;
2. optionally one all-gather along ``B`` of the output.

Everything else (top-k, grouping, convs, aggregation) is rank-local.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist
from torch import nn


def _refuse_flagged_grouping(block_index=None) -> None:
    """The reference's TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING grouping variants rank candidate groups by batch
    means of the amplitudes (:374-378, :394-437).  A sharded run exchanges only the [F] spectrum sums, so each rank
    would rank on its own rows and the ranks could keep different groups: refuse instead of diverging silently."""
    from .grouping import _resolve_log_binning_base, _resolve_scheduled_int

    if (_resolve_scheduled_int(os.getenv("TIMES_PERIOD_MAX_UNIQ"), block_index) or
            _resolve_log_binning_base(os.getenv("TIMES_PERIOD_BINNING"), block_index)):
        raise NotImplementedError("TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING are not supported with a batch-sharded "
                                  "selector: unset them or run unsharded")


class IpcExchange:
    """SURVEY section 8e step 2 without a collective: every rank owns one exchange buffer in its GPU's memory, maps the
    other ranks' buffers once (``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle``, handles traded through the process
    group), and from then on each block call's partial sums are plain stores into the peers' buffers issued by the
    selector's own kernel, with a sequence word behind them (``include/flowtimes.h``, ``FtnExchange``).  Attach it with
    ``ShardedTimesBlock(block, group, exchange=IpcExchange(group, device))``; ranks must call in lockstep."""

    _IPC_HANDLE_BYTES = 64

    def __init__(self, group, device: torch.device, f_cap: int = 1024) -> None:
        import ctypes as C

        from . import lib as _lib

        self.group = group if group is not None else dist.group.WORLD
        self.world, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if self.world > _lib.FTN_XCHG_MAXWORLD:
            raise ValueError(f"IpcExchange supports up to {_lib.FTN_XCHG_MAXWORLD} ranks")
        self.device = torch.device(device)
        lib = _lib.load()
        nbytes = lib.ftn_exchange_bytes(self.world, int(f_cap))
        if nbytes == 0:
            raise ValueError(f"ftn_exchange_bytes rejected world={self.world} F_cap={f_cap}")
        self._C, self._lib = C, lib
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)                      # the HIP context of this device exists
        own = C.c_void_p()
        handle = C.create_string_buffer(self._IPC_HANDLE_BYTES)
        _lib.check(lib.ftn_exchange_alloc(self.world, int(f_cap), C.byref(own), handle), "ftn_exchange_alloc")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw), group=self.group)
        self._own, self._mapped = own, []
        self.x = _lib.FtnExchange()
        self.x.world, self.x.rank, self.x.F_cap, self.x.seq = self.world, self.rank, int(f_cap), 0
        for r, raw in enumerate(handles):
            if r == self.rank:
                self.x.slots[r] = own.value
                continue
            peer = C.c_void_p()
            _lib.check(lib.ftn_exchange_open(C.create_string_buffer(raw, self._IPC_HANDLE_BYTES), C.byref(peer)),
                       f"ftn_exchange_open(rank {r})")
            self._mapped.append(peer)
            self.x.slots[r] = peer.value
        dist.barrier(group=self.group)                          # every rank has zeroed and mapped before the first call

    def next_call(self, F: int):
        """The struct pointer for one exchange (bumps the sequence number: every rank must make the same calls)."""
        if F > self.x.F_cap:
            raise ValueError(f"IpcExchange: F={F} exceeds F_cap={self.x.F_cap}")
        self.x.seq += 1
        return self._C.byref(self.x)

    def check(self) -> None:
        """Synchronises; raises if a peer's sums did not arrive within the kernel's bounded wait."""
        from . import lib as _lib
        from . import runtime

        rc = _lib.load().ftn_exchange_error(self._C.byref(self.x), runtime._stream(self.device))
        if rc != 0:
            raise RuntimeError("IpcExchange: a peer's partial sums did not arrive (timeout)" if rc == 1
                               else f"ftn_exchange_error rc={rc}")

    def close(self) -> None:
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)                          # nobody still writes into a buffer that is going away
        for peer in self._mapped:
            self._lib.ftn_exchange_close(peer)
        self._mapped = []
        if self._own is not None:
            self._lib.ftn_exchange_free(self._own)
            self._own = None


def gather_batch(y_local: torch.Tensor, group=None, async_op: bool = False):
    """All-gather equal-sized shards along dim 0 -> ``[world*B_local, ...]`` on every rank.

    ``async_op=True`` returns ``(out, work)``: the collective runs on the backend's own stream
    (RCCL over xGMI) and overlaps whatever the caller enqueues next; ``work.wait()`` makes the
    current stream wait for it.  ``out`` must not be read before that."""
    world = dist.get_world_size(group)
    if world == 1 and os.environ.get("FTN_BENCH_FORCE_DIST") != "1":
        return (y_local, None) if async_op else y_local
    y_local = y_local.contiguous()
    out = y_local.new_empty((world * y_local.shape[0],) + tuple(y_local.shape[1:]))
    if y_local.is_cuda and dist.get_backend(group) != "gloo":
        work = dist.all_gather_into_tensor(out, y_local, group=group, async_op=async_op)
    else:  # gloo (CPU tests)
        parts = list(out.chunk(world, dim=0))
        work = dist.all_gather(parts, y_local, group=group, async_op=async_op)
    return (out, work) if async_op else out


class ShardedTimesBlock(nn.Module):
    """Wraps a ``TimesBlock`` (with an ``FFTPeriodSelector``) for batch-sharded use.

    ``forward(x_local)`` returns this rank's rows (``gather=False``) or the
    re-assembled global batch (``gather=True``).  Shards must be equally sized.
    """

    def __init__(self, block: nn.Module, group=None, exchange: Optional[IpcExchange] = None) -> None:
        super().__init__()
        self.block = block
        self.group = group
        self.exchange = exchange                                 # None: all-gather through torch.distributed (RCCL / gloo)
        sel = block.period_selector
        if sel is None or not hasattr(sel, "shard_group"):
            raise ValueError("ShardedTimesBlock needs a block with a native FFTPeriodSelector")

    def forward(self, x_local: torch.Tensor, gather=True):
        """``gather``: ``False`` -> this rank's rows; ``True`` -> the re-assembled global batch;
        ``"async"`` -> ``(out, work)`` with the all-gather still in flight (see ``gather_batch``),
        which lets a serving loop overlap step i's output exchange with step i+1's compute."""
        sel = self.block.period_selector
        grp = self.group if self.group is not None else dist.group.WORLD
        if dist.get_world_size(grp) > 1:
            _refuse_flagged_grouping(getattr(self.block, "block_index", None))
        prev, prev_x = sel.shard_group, sel.shard_exchange
        sel.shard_group = grp if (dist.get_world_size(grp) > 1 or os.environ.get("FTN_BENCH_FORCE_DIST") == "1") else None
        sel.shard_exchange = self.exchange if (self.exchange is not None and x_local.is_cuda) else None
        try:
            y = self.block(x_local)
        finally:
            sel.shard_group, sel.shard_exchange = prev, prev_x
        if gather == "async":
            return gather_batch(y, grp, async_op=True)
        return gather_batch(y, grp) if gather else y


class ShardedTimesNet(nn.Module):
    """Batch-sharded whole model (P2): the model's blocks share one ``FFTPeriodSelector``, so
    pointing its ``shard_group`` at the process group makes every block exchange its ``[F]`` partial sums;
    everything else in ``TimesNet.forward`` is row-wise or per-series and needs no communication.
    ``forward`` returns this rank's ``(rate, dispersion)`` rows, or the all-gathered ones with ``gather=True``."""

    def __init__(self, model: nn.Module, group=None) -> None:
        super().__init__()
        self.model = model
        self.group = group
        if not hasattr(model.period_selector, "shard_group"):
            raise ValueError("ShardedTimesNet needs the mirror TimesNet (native FFTPeriodSelector)")

    def forward(self, x_local: torch.Tensor, gather: bool = False, **kwargs):
        sel = self.model.period_selector
        grp = self.group if self.group is not None else dist.group.WORLD
        if dist.get_world_size(grp) > 1:
            for blk in self.model.blocks:
                _refuse_flagged_grouping(getattr(blk, "block_index", None))
        prev = sel.shard_group
        sel.shard_group = grp if (dist.get_world_size(grp) > 1 or os.environ.get("FTN_BENCH_FORCE_DIST") == "1") else None
        try:
            rate, disp = self.model(x_local, **kwargs)
        finally:
            sel.shard_group = prev
        if gather:
            return gather_batch(rate, grp), gather_batch(disp, grp)
        return rate, disp
