"""HIP-graph replay of an inference forward.

The HIP path of this package never reads device results on the host while a forward is being
enqueued (the period descriptor, group weights and tile counts stay on the device and every grid is
sized for the worst case), so a whole ``TimesBlock`` or ``TimesNet`` forward - a few dozen launches,
most of them short - can be captured once and replayed with a single ``hipGraphLaunch``.  That
removes the per-launch host cost that otherwise leaves the GPU idle between the small front-end
kernels of the model shell.

Only the one host read the reference semantics require survives: ``TimesNet``'s "rate / dispersion
must be finite and > 0" check, deferred to after the replay (``TimesNet.check_outputs``).
"""
from __future__ import annotations

from typing import Any, Dict, Sequence, Tuple

import torch
from torch import nn


class GraphedForward:
    """``g = GraphedForward(module, x, **kwargs)`` captures ``module(x, **kwargs)`` (eval mode, no
    autograd) on ``x``'s device; ``g(x_new, **kwargs_new)`` copies the new tensors into the captured
    input buffers, replays, and returns the captured output tensors (overwritten by the next call).
    Shapes, dtypes and non-tensor arguments are frozen at capture time."""

    def __init__(self, module: nn.Module, *inputs: Any, warmup: int = 2, **kwargs: Any) -> None:
        tensors = [t for t in list(inputs) + list(kwargs.values()) if isinstance(t, torch.Tensor)]
        if not tensors or not tensors[0].is_cuda:
            raise ValueError("GraphedForward needs ROCm tensors")
        self.module = module.eval()
        self.device = tensors[0].device
        self._in = tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in inputs)
        self._kw: Dict[str, Any] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
        self._has_checks = hasattr(module, "check_outputs")
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(max(1, warmup)):            # lazy builds, weight packing, tables, workspace
                module(*self._in, **self._kw)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        if self._has_checks:
            module._defer_checks = True
        try:
            with torch.inference_mode(), torch.cuda.graph(self.graph):
                self._out = module(*self._in, **self._kw)
        finally:
            if self._has_checks:
                module._defer_checks = False
        self._pending = getattr(module, "_pending_bad", None)
        # the captured launches have the packed weight blobs' addresses baked in: keep them alive even if a block
        # repacks later (new weights need a new capture)
        self._packs = [m._pack for m in module.modules() if getattr(m, "_pack", None) is not None]

    @staticmethod
    def _refill(dst: Any, src: Any) -> None:
        if isinstance(dst, torch.Tensor):
            if not isinstance(src, torch.Tensor) or src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError("GraphedForward: input does not match the captured shape/dtype")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        elif dst is not src and dst != src:
            raise ValueError("GraphedForward: non-tensor arguments are frozen at capture time")

    @property
    def inputs(self) -> Tuple[Any, ...]:
        """The captured input buffers; writing into them directly saves the copy in ``__call__``."""
        return self._in

    def replay(self, check: bool = True):
        self.graph.replay()
        if check and self._has_checks and self._pending is not None:
            self.module._pending_bad = self._pending
            self.module.check_outputs()
        return self._out

    def __call__(self, *inputs: Any, check: bool = True, **kwargs: Any):
        if len(inputs) != len(self._in) or set(kwargs) != set(self._kw):
            raise ValueError("GraphedForward: call signature differs from the captured one")
        for dst, src in zip(self._in, inputs):
            self._refill(dst, src)
        for k, src in kwargs.items():
            self._refill(self._kw[k], src)
        return self.replay(check)
