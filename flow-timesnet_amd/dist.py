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

from typing import Optional

import torch
import torch.distributed as dist
from torch import nn


def gather_batch(y_local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather equal-sized shards along dim 0 -> ``[world*B_local, ...]`` on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    y_local = y_local.contiguous()
    out = y_local.new_empty((world * y_local.shape[0],) + tuple(y_local.shape[1:]))
    if y_local.is_cuda:
        dist.all_gather_into_tensor(out, y_local, group=group)
    else:  # gloo (CPU tests)
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, y_local, group=group)
    return out


class ShardedTimesBlock(nn.Module):
    """Wraps a ``TimesBlock`` (with an ``FFTPeriodSelector``) for batch-sharded use.

    ``forward(x_local)`` returns this rank's rows (``gather=False``) or the
    re-assembled global batch (``gather=True``).  Shards must be equally sized.
    """

    def __init__(self, block: nn.Module, group=None) -> None:
        super().__init__()
        self.block = block
        self.group = group
        sel = block.period_selector
        if sel is None or not hasattr(sel, "shard_group"):
            raise ValueError("ShardedTimesBlock needs a block with a native FFTPeriodSelector")

    def forward(self, x_local: torch.Tensor, gather: bool = True) -> torch.Tensor:
        sel = self.block.period_selector
        grp = self.group if self.group is not None else dist.group.WORLD
        prev = sel.shard_group
        sel.shard_group = grp if dist.get_world_size(grp) > 1 else None
        try:
            y = self.block(x_local)
        finally:
            sel.shard_group = prev
        return gather_batch(y, grp) if gather else y
