"""Drop-in mirrors of the hot-path modules of the reference's
``timesnet_forecast/models/timesnet.py`` — same class names, constructor
signatures, observable attributes and ``state_dict`` keys — whose forward runs on
hand-written HIP kernels when the input lives on a ROCm device.

Backends
--------
``hip``    input is a CUDA(ROCm) tensor, autograd is not recording, and
           ``block.inception`` is the standard ``Sequential(InceptionBlock, act,
           InceptionBlock)``: everything runs in ``libflowtimes_hip.so``.  A missing
           library is an error, never a silent fallback.
``torch``  anything else (CPU tensors, training with autograd/dropout, an injected
           ``inception`` module as in the reference's tests): stock torch ops
           composed here the way the reference composes them.

``block._last_backend`` records which one ran; GPU parity tests assert ``"hip"``.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import synth
from ..grouping import PeriodGrouper, PeriodGroupResult  # noqa: F401  (re-exported, reference API)
from ..lib import FTN_KMAX


def _safe_param_dtype(dtype: torch.dtype) -> torch.dtype:
    """fp16/bf16 activations keep fp32 parameters (reference :14-34)."""
    return torch.float32 if dtype in (torch.float16, torch.bfloat16) else dtype


def _env_on(name: str) -> bool:
    v = os.getenv(name)
    return bool(v) and v.strip().lower() not in {"0", "false", "off"}


def _hip_eligible(x: torch.Tensor, *modules) -> bool:
    """The HIP kernels are forward-only: they run when the input lives on a ROCm device and autograd has
    nothing to record - grad mode off, or neither the input nor any parameter of ``modules`` requires
    grad.  (A frozen input in front of trainable blocks must still take the torch path, or the block
    weights would silently receive no gradient.)"""
    if not x.is_cuda:
        return False
    if not torch.is_grad_enabled():
        return True
    if x.requires_grad:
        return False
    return not any(p.requires_grad for m in modules if m is not None for p in m.parameters())


# =========================================================================
# FFTPeriodSelector                                   reference :52-159
# =========================================================================
class FFTPeriodSelector(nn.Module):
    """Shared dominant-period selector: rFFT amplitude, lower median over
    channels, mean over the batch, top-k, frequency -> period."""

    def __init__(self, k_periods: int, pmax: int, min_period_threshold: int = 1) -> None:
        super().__init__()
        self.k = int(max(0, k_periods))
        self.pmax = int(max(1, pmax))
        self.min_period_threshold = int(min(self.pmax, max(1, int(min_period_threshold))))
        # batch-sharded multi-GPU: when set, the [F] batch sums are exchanged over this
        # process group so every rank selects identical periods
        self.shard_group = None
        # dist.IpcExchange: the partial sums travel as peer writes into IPC-mapped buffers instead of a collective
        self.shard_exchange = None
        self._pending = None
        self._lfi = torch.zeros(0, dtype=torch.long)
        self._lsp = torch.zeros(0, dtype=torch.long)

    # The reference stores these as plain tensors after each call (:86-87,156-157).
    # Here they are materialised lazily so the HIP path never synchronises on its own.
    def _materialise(self) -> None:
        sel = self._pending
        if sel is not None:
            self._pending = None
            d = sel.host()
            dev = sel.desc.device
            n = int(d.n_sel)
            self._lfi = torch.tensor(list(d.sel_freq[:n]), dtype=torch.long, device=dev)
            self._lsp = torch.tensor(list(d.sel_period[:n]), dtype=torch.long, device=dev)

    @property
    def last_frequency_indices(self) -> torch.Tensor:
        self._materialise()
        return self._lfi

    @last_frequency_indices.setter
    def last_frequency_indices(self, v: torch.Tensor) -> None:
        self._pending = None
        self._lfi = v

    @property
    def last_selected_periods(self) -> torch.Tensor:
        self._materialise()
        return self._lsp

    @last_selected_periods.setter
    def last_selected_periods(self, v: torch.Tensor) -> None:
        self._pending = None
        self._lsp = v

    # ---- HIP: no host sync, result stays on the device --------------------
    def _degenerate(self, x: torch.Tensor) -> bool:
        B, L, C = x.shape
        return self.k <= 0 or L <= 1 or C <= 0 or B <= 0 or min(self.pmax, max(1, L - 1)) < self.min_period_threshold

    def select_device(self, x: torch.Tensor, act_dtype: Optional[torch.dtype] = None,
                      max_unique: Optional[int] = None, log_base: Optional[float] = None, stage_a=None):
        """Run S1-S5 on the device; returns a ``runtime.Selection`` (or ``None`` when
        the selector is degenerate, reference :89-90,140-142).  ``act_dtype`` (default: ``x.dtype``) is the
        caller's activation dtype: for bf16 / fp16 the reference's roundings of scores, amplitudes and
        softmax weights are applied (:124-159, :1000-1009).  ``stage_a=(plan, wblob[, range_flag])`` of the TimesBlock
        that will consume the selection lets its first stage share the finalize launch (``runtime.finalize``)."""
        from .. import runtime

        if x.ndim != 3:
            raise ValueError("FFTPeriodSelector expects input shaped [B, L, C]")
        if self._degenerate(x):
            self.last_frequency_indices = torch.zeros(0, dtype=torch.long, device=x.device)
            self.last_selected_periods = torch.zeros(0, dtype=torch.long, device=x.device)
            return None
        if self.k > FTN_KMAX:
            raise ValueError(f"k_periods={self.k} exceeds the native limit FTN_KMAX={FTN_KMAX}")
        B, L, _ = x.shape
        adt = runtime.ACT_DTYPE.get(act_dtype if act_dtype is not None else x.dtype, 0)
        xf = x.detach()
        if xf.dtype != torch.float32:
            xf = xf.float()                                   # FFT is always >= fp32 (:92-94)
        xf = xf.contiguous()
        xch = None
        if self.shard_group is not None and self.shard_exchange is not None:
            xch = self.shard_exchange.next_call(L // 2 + 1)        # this call's sequence number, same on every rank
        med, psum = runtime.spectrum(xf, xch)
        b_total = B
        pre = None
        if xch is not None:
            # k_colsum has stored this rank's sums into every peer's buffer; the finalize workgroup waits for the
            # peers' stores (bounded) while stage A runs beside it - no collective, nothing else for the host to do
            if (max_unique or 0) > 0 or (log_base or 0.0) > 1.0:
                raise NotImplementedError("TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING are not supported with a "
                                          "batch-sharded selector (shard_group): unset them or run unsharded")
            b_total = B * self.shard_exchange.world
        elif self.shard_group is not None:
            import torch.distributed as dist

            if (max_unique or 0) > 0 or (log_base or 0.0) > 1.0:
                # those variants rank candidate groups by batch means of the amplitudes (:374-378, :394-437); only the
                # [F] sums are exchanged, so every rank would rank on its own rows and could keep different groups
                raise NotImplementedError("TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING are not supported with a "
                                          "batch-sharded selector (shard_group): unset them or run unsharded")
            world = dist.get_world_size(self.shard_group)
            parts = torch.empty(world, psum.numel(), dtype=psum.dtype, device=psum.device)
            # the block's stage A needs x only: it runs while the partial sums travel
            overlap = stage_a is not None and runtime.fuse_stage_a(stage_a[0])
            if dist.get_backend(self.shard_group) == "gloo":      # CPU-side rehearsal of the exchange
                if overlap:
                    pre = runtime.stage_a_only(xf, stage_a[0], stage_a[1], self.k, self.pmax, self.min_period_threshold,
                                               stage_a[2] if len(stage_a) > 2 else None)
                dist.all_gather(list(parts.unbind(0)), psum, group=self.shard_group)
            else:                                                  # RCCL
                work = dist.all_gather_into_tensor(parts, psum, group=self.shard_group, async_op=True)
                if overlap:                                        # enqueued behind the spectrum, beside the exchange
                    pre = runtime.stage_a_only(xf, stage_a[0], stage_a[1], self.k, self.pmax, self.min_period_threshold,
                                               stage_a[2] if len(stage_a) > 2 else None)
                work.wait()                                        # the compute stream waits; the host does not
            b_total = B * world          # equal shards (no host sync to learn otherwise)
            psum = parts
        sel = runtime.finalize(psum, b_total, med, L, self.k, self.pmax, self.min_period_threshold, adt,
                               max_unique or 0, log_base or 0.0,
                               stage_a=None if stage_a is None else (xf,) + tuple(stage_a), pre=pre, xch=xch)
        self._pending = sel
        return sel

    # ---- torch ops (CPU tensors) --------------------------------------------
    def _forward_torch(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        B, L, C = x.shape
        dev, dtype = x.device, x.dtype
        empty_idx = torch.zeros(0, dtype=torch.long, device=dev)
        empty_amp = torch.zeros(B, 0, dtype=dtype, device=dev)
        self.last_frequency_indices = empty_idx
        self.last_selected_periods = empty_idx
        if self._degenerate(x):
            return empty_idx, empty_amp
        xf = x.float() if dtype in (torch.float16, torch.bfloat16) else x
        med = torch.fft.rfft(xf, dim=1).abs().median(dim=2).values            # [B, F]
        if self.shard_group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(self.shard_group)
            part = med.sum(dim=0, dtype=torch.float64)
            parts = [torch.empty_like(part) for _ in range(world)]
            dist.all_gather(parts, part, group=self.shard_group)
            cnt = torch.tensor([B], dtype=torch.int64)
            cnts = [torch.empty_like(cnt) for _ in range(world)]
            dist.all_gather(cnts, cnt, group=self.shard_group)
            tot = sum(int(c.item()) for c in cnts)
            acc = torch.zeros_like(part)
            for p in parts:                                                    # fixed rank order
                acc = acc + p
            mean = (acc / tot).to(med.dtype)
        else:
            mean = med.mean(dim=0)
        nbins = mean.numel()
        if nbins <= 1:
            return empty_idx, empty_amp
        mean = mean.to(dtype).clone()
        mean[0] = float("-inf")
        k = min(self.k, nbins - 1)
        if k <= 0:
            return empty_idx, empty_amp
        pen = torch.log1p(torch.arange(nbins, device=dev, dtype=torch.float32))
        scores = mean - 1e-8 * pen.to(dtype)
        idx = torch.topk(scores, k=k, largest=True).indices.clamp_min(1)
        amps = med.gather(1, idx.view(1, -1).expand(B, -1))
        hi = min(self.pmax, max(1, L - 1))
        periods = torch.clamp((L + idx - 1) // idx, min=self.min_period_threshold, max=hi)
        keep = ((L + periods - 1) // periods) >= 2
        if not bool(keep.any()):
            return empty_idx, empty_amp
        self.last_frequency_indices = idx[keep]
        self.last_selected_periods = periods[keep]
        return periods[keep], amps[:, keep].to(dtype)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """``x[B, L, C]`` -> (``periods[K']`` long, ``amplitudes[B, K']`` in x.dtype)."""
        if x.ndim != 3:
            raise ValueError("FFTPeriodSelector expects input shaped [B, L, C]")
        if not _hip_eligible(x):
            return self._forward_torch(x)
        sel = self.select_device(x)
        if sel is None:
            return (torch.zeros(0, dtype=torch.long, device=x.device),
                    torch.zeros(x.shape[0], 0, dtype=x.dtype, device=x.device))
        n = int(sel.host().n_sel)          # data-dependent output shape: one sync, as in the reference
        return self.last_selected_periods, sel.amps[:, :n].to(x.dtype)


# =========================================================================
# Inception                                           reference :560-654
# =========================================================================
class InceptionBranch(nn.Module):
    """One branch: a single conv (ratio ~ 1) or 1x1 -> kxk -> 1x1, no activations."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: Tuple[int, int], bottleneck_ratio: float) -> None:
        super().__init__()
        if bottleneck_ratio <= 0:
            raise ValueError("bottleneck_ratio must be a positive value")
        kh, kw = kernel_size
        pad = (max(kh // 2, 0), max(kw // 2, 0))
        mid = synth.bottleneck_mid(in_ch, out_ch, bottleneck_ratio)
        if mid is None:
            layers = [nn.Conv2d(in_ch, out_ch, kernel_size=(kh, kw), padding=pad)]
        else:
            layers = [nn.Conv2d(in_ch, mid, kernel_size=1),
                      nn.Conv2d(mid, mid, kernel_size=(kh, kw), padding=pad),
                      nn.Conv2d(mid, out_ch, kernel_size=1)]
        self.branch = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.branch(x)


class InceptionBlock(nn.Module):
    """cat(branches) -> proj 1x1 -> act -> dropout -> + res_proj(x)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_set, dropout: float, act: str,
                 bottleneck_ratio: float = 1.0) -> None:
        super().__init__()
        kernels = synth.parse_kernel_set(kernel_set)
        self.paths = nn.ModuleList(
            [InceptionBranch(in_ch, out_ch, k, bottleneck_ratio) for k in kernels]
        )
        self.proj = nn.Conv2d(out_ch * len(kernels), out_ch, kernel_size=1)
        self.res_proj = nn.Conv2d(in_ch, out_ch, kernel_size=1) if in_ch != out_ch else nn.Identity()
        self.dropout = nn.Dropout(dropout)
        self.act = nn.ReLU() if act.lower() == "relu" else nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        z = self.proj(torch.cat([p(x) for p in self.paths], dim=1))
        return self.dropout(self.act(z)) + self.res_proj(x)


# =========================================================================
# TimesBlock                                          reference :657-1101
# =========================================================================
class TimesBlock(nn.Module):
    """``x[B, L, C] -> x + sum_g w[b,g] * (inception(fold_g(x)) - fold_g(x))``."""

    def __init__(self, d_model: Optional[int], kernel_set, dropout: float, activation: str,
                 d_ff: Optional[int] = None, bottleneck_ratio: float = 1.0) -> None:
        super().__init__()
        self._configured_d_model = int(d_model) if d_model is not None else None
        self._configured_d_ff = None if d_ff is None else int(d_ff)
        if self._configured_d_ff is not None and self._configured_d_ff <= 0:
            raise ValueError("d_ff must be a positive integer")
        self.d_model: Optional[int] = None
        self.d_ff: Optional[int] = None
        self.bottleneck_ratio = float(bottleneck_ratio)
        if self.bottleneck_ratio <= 0:
            raise ValueError("bottleneck_ratio must be a positive value")
        self._activation_name = "relu" if activation.lower() == "relu" else "gelu"
        self._kernel_spec = synth.parse_kernel_set(kernel_set)
        self._dropout = float(dropout)
        self.inception: Optional[nn.Module] = None
        if self._configured_d_model is not None:
            self._build_layers(self._configured_d_model, torch.device("cpu"), torch.get_default_dtype())
        # injected by the owner (TimesNet) after construction, reference :711-713
        self.period_selector: Optional[nn.Module] = None
        self._period_calls = 0
        self._vec_calls = 0
        self.block_index: Optional[int] = None
        self._last_raw_period_count = 0
        self._last_valid_period_count = 0
        self._last_group_count = 0
        self._last_loop_iterations = 0
        self._last_backend: Optional[str] = None
        self._hip_calls = 0
        self._lazy_sel = None
        self._pack_key = None
        self._pack = None
        self._packs = {}
        # conv arithmetic of the HIP backend: None = pack.default_engine() ("f16x2" unless FLOWTIMES_ENGINE says
        # otherwise); "bf16x3" = three bf16 pieces, full fp32 exponent range; "f32" = exact fp32 MFMA; "bf16" = plain bf16
        self.engine: Optional[str] = None
        # f16x2 range guard (include/flowtimes.h, ABI 9): the kernels flag values that do not fit fp16 pieces; the
        # flag of a call is looked at once its completion event has fired - at the next call of this block, by
        # check_range(), or when _last_engine is read - and a flagged call is repeated on bf16x3 INTO THE SAME output
        self._range_slots: list = []          # free (flag, event) pairs
        self._range_pending: list = []        # calls in flight: (flag, event, x, y, post_norm)
        self._range_dev_flag = None           # device-memory flag used while a HIP graph is being captured
        self._last_engine_used: Optional[str] = None
        self._range_fallbacks = 0

    # ---- construction ------------------------------------------------------
    def _build_layers(self, channels: int, device: torch.device, dtype: torch.dtype) -> None:
        if channels <= 0:
            raise ValueError("TimesBlock requires a positive channel count")
        self.d_model = int(channels)
        self.d_ff = int(self._configured_d_ff if self._configured_d_ff is not None else self.d_model)
        act = nn.ReLU() if self._activation_name == "relu" else nn.GELU()
        mk = lambda i, o: InceptionBlock(i, o, self._kernel_spec, self._dropout, self._activation_name,
                                         self.bottleneck_ratio)
        self.inception = nn.Sequential(mk(self.d_model, self.d_ff), act, mk(self.d_ff, self.d_model)).to(
            device=device, dtype=_safe_param_dtype(dtype))

    def _standard_inception(self) -> bool:
        inc = self.inception
        return (isinstance(inc, nn.Sequential) and len(inc) == 3 and isinstance(inc[0], InceptionBlock)
                and isinstance(inc[2], InceptionBlock))

    # ---- lazily materialised counters (HIP path keeps them on the device) ---
    def _sync_counters(self) -> None:
        sel = self.__dict__.get("_lazy_sel")
        if sel is not None:
            self.__dict__["_lazy_sel"] = None
            d = sel.host()
            n = int(d.n_sel)
            self.__dict__["_c_raw"] = n
            self.__dict__["_c_valid"] = sum(1 for j in range(n) if d.sel_group[j] >= 0)
            self.__dict__["_c_groups"] = int(d.n_groups)

    def _counter(name):  # noqa: N805
        def get(self):
            self._sync_counters()
            return self.__dict__.get(name, 0)

        def set_(self, v):
            self.__dict__[name] = v

        return property(get, set_)

    _last_raw_period_count = _counter("_c_raw")
    _last_valid_period_count = _counter("_c_valid")
    _last_group_count = _counter("_c_groups")
    del _counter

    # ---- packed weights for the HIP kernels ---------------------------------
    def _packed(self, device: torch.device, engine: Optional[str] = None):
        from .. import pack

        params = list(self.inception.parameters())
        engine = engine or getattr(self, "engine", None) or pack.default_engine()
        # inference tensors (lazy build under torch.inference_mode) carry no version counter
        key = (str(device),) + tuple(
            (p.data_ptr(), -1 if p.is_inference() else p._version) for p in params)
        if self._pack_key != key:                               # new weights / device: every engine's blob is stale
            self._packs = {}
            self._pack_key = key
        if engine not in self._packs:
            sd = {k: v.detach().float().cpu().numpy() for k, v in self.inception.state_dict().items()}
            blob, plan = pack.pack_inception(sd, self.d_model, self.d_ff, self._kernel_spec,
                                             self.bottleneck_ratio, self._activation_name, engine)
            self._packs[engine] = (torch.from_numpy(blob).to(device), plan)
        self._pack = self._packs[engine]
        return self._pack

    def invalidate_pack(self) -> None:
        """Forget the packed weight blob.  The cache is keyed on the parameters' storage pointers and version
        counters, which catches ``load_state_dict`` / optimiser steps / ``.to()``; inference tensors carry no
        version counter, so after an IN-PLACE update of parameters created under ``torch.inference_mode`` call
        this (or replace the parameters) before the next forward.  Graphs captured earlier keep the blob they
        were captured with (``graph.GraphedForward`` holds a reference)."""
        self._pack_key = None
        self._pack = None
        self._packs = {}

    # ---- f16x2 range guard -------------------------------------------------------
    def _engine_name(self) -> str:
        from .. import pack

        return getattr(self, "engine", None) or pack.default_engine()

    def _resolve_range(self, block: bool) -> None:
        """Look at the range flags of finished f16x2 calls (all pending calls when ``block``); a flagged call is
        repeated on the bf16x3 engine into the output tensor it returned."""
        while self._range_pending:
            flag, ev, x, y, post_norm = self._range_pending[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            self._range_pending.pop(0)
            tripped = int(flag.item()) != 0                      # pinned host word: a plain read
            if tripped:
                flag.zero_()
            self._range_slots.append((flag, ev))
            self._last_engine_used = "f16x2"
            if tripped:
                import warnings

                warnings.warn("TimesBlock: a value left the fp16 range of engine f16x2 (|v| >= 65504 or not finite); "
                              "the call was repeated on engine bf16x3", RuntimeWarning, stacklevel=3)
                self._range_fallbacks += 1
                y2 = self._forward_hip(x, post_norm, engine="bf16x3")
                y.copy_(y2 if y2.dtype == y.dtype else y2.to(y.dtype))
                self._last_engine_used = "bf16x3"

    def check_range(self) -> Optional[str]:
        """Wait for this block's outstanding HIP calls and repair any whose values left the f16x2 engine's fp16
        range (see ``_resolve_range``); returns the engine the last call finally ran on."""
        self._resolve_range(block=True)
        flag = self._range_dev_flag
        if flag is not None and int(flag.item()) != 0:           # set during a captured / deferred forward
            raise FloatingPointError("TimesBlock: a value left the fp16 range of engine f16x2 inside a captured "
                                     "forward; set block.engine = 'bf16x3' and capture again")
        return self._last_engine_used

    @property
    def _last_engine(self) -> Optional[str]:
        return self.check_range()

    # ---- forward ---------------------------------------------------------------
    def forward(self, x: torch.Tensor, post_norm: Optional[nn.LayerNorm] = None) -> torch.Tensor:
        """``post_norm`` (extension, used by the TimesNet shell in eval mode): also apply the model's
        per-block ``post_norm(x + (block(x) - x))`` (reference :2050-2058) - inside the last HIP kernel
        on the HIP backend, as plain torch ops otherwise."""
        if x.ndim != 3:
            raise ValueError("TimesBlock expects input shaped [B, L, d_model]")
        if self.period_selector is None:
            raise RuntimeError("TimesBlock.period_selector has not been set")
        self._period_calls += 1
        if self.inception is None:
            if self._configured_d_model is not None and x.size(-1) != self._configured_d_model:
                raise ValueError("Configured d_model does not match the incoming channel dimension")
            self._build_layers(x.size(-1), x.device, x.dtype)
        else:
            self.inception = self.inception.to(device=x.device, dtype=_safe_param_dtype(x.dtype))
            if self.d_model is not None and x.size(-1) != self.d_model:
                raise ValueError("Number of channels changed between calls")

        use_hip = (self._standard_inception() and not (self.training and self._dropout > 0.0)
                   and _hip_eligible(x, self.inception) and self._within_native_limits())
        fused = (post_norm is not None and use_hip and _affine_layernorm(post_norm, x.size(-1))
                 and _hip_eligible(x, post_norm) and x.dtype == torch.float32)
        if not use_hip:
            self._last_backend = "torch"
            new = self._forward_torch(x)
        else:
            self._last_backend = "hip"
            self._hip_calls += 1
            new = self._forward_hip(x, post_norm if fused else None)
        if post_norm is not None and not fused:
            new = _layer_norm_fp32(post_norm, x + (new - x))
        return new

    def _within_native_limits(self) -> bool:
        """The HIP kernels hold at most FTN_KMAX period candidates and FTN_MAXBR kernels per InceptionBlock; the
        reference has no such limits (:55-62, :622-633), so anything beyond them runs on the torch backend."""
        from ..lib import FTN_MAXBR

        sel = self.period_selector
        k = getattr(sel, "k", None) if type(sel) is FFTPeriodSelector else None
        return len(self._kernel_spec) <= FTN_MAXBR and (k is None or int(k) <= FTN_KMAX)

    # ---- HIP backend -----------------------------------------------------------
    def _forward_hip(self, x: torch.Tensor, post_norm: Optional[nn.LayerNorm] = None,
                     engine: Optional[str] = None) -> torch.Tensor:
        from .. import lib, runtime

        engine = engine or self._engine_name()
        guarded = engine == "f16x2" and os.getenv("FTN_RANGE_GUARD", "1") != "0"
        range_flag = slot_ev = None
        if guarded:
            if torch.cuda.is_current_stream_capturing():
                # a captured forward cannot record host events: it sets a device word (zeroed by a captured fill at
                # every replay) that check_range() / TimesNet.check_outputs() read after the replay
                self._range_dev_flag = range_flag = torch.zeros(1, dtype=torch.int32, device=x.device)
            else:
                self._range_dev_flag = None
                self._resolve_range(block=False)
                if len(self._range_pending) >= 8:              # the host runs far ahead: wait for the oldest call only
                    self._range_pending[0][1].synchronize()
                    self._resolve_range(block=False)
                range_flag, slot_ev = self._range_slots.pop() if self._range_slots else (
                    runtime.new_range_flag(x.device), torch.cuda.Event())

        norm = None
        if post_norm is not None:
            norm = (post_norm.weight.detach().float().contiguous(), post_norm.bias.detach().float().contiguous(),
                    post_norm.eps)

        def unchanged():
            # the reference returns x itself (:796-797); the shell then normalises x + (x - x)
            if slot_ev is not None:
                self._range_slots.append((range_flag, slot_ev))
            if norm is None:
                return x
            xc = x.detach().float().contiguous()
            y = runtime.residual_layernorm(xc, xc, *norm)
            return y if y.dtype == x.dtype else y.to(x.dtype)

        B, L, _ = x.shape
        sel_mod = self.period_selector
        native = type(sel_mod) is FFTPeriodSelector
        # TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING (per-depth schedules resolved here, :320-325) are grouping
        # variants of the device finalize kernel: no host round trip on the native path
        from ..grouping import _resolve_log_binning_base, _resolve_scheduled_int
        max_unique = _resolve_scheduled_int(os.getenv("TIMES_PERIOD_MAX_UNIQ"), self.block_index)
        log_base = _resolve_log_binning_base(os.getenv("TIMES_PERIOD_BINNING"), self.block_index)
        xf = x.detach()
        if xf.dtype != torch.float32:
            xf = xf.float()
        xf = xf.contiguous()
        adt = runtime.ACT_DTYPE.get(x.dtype, 0)
        if adt != 0:
            norm = None                        # half inputs: the shell's LayerNorm runs outside (see forward)
        wblob, plan = self._packed(x.device, engine)
        if native:
            sel = sel_mod.select_device(xf, act_dtype=x.dtype, max_unique=max_unique, log_base=log_base,
                                        stage_a=(plan, wblob, range_flag))
            if sel is None:                                            # reference :796-797
                self._last_raw_period_count = self._last_valid_period_count = self._last_group_count = 0
                return unchanged()
            self._lazy_sel = sel                                       # counters resolve on access
        else:
            # foreign selector (reference tests inject stubs, :711-713) or env-flag
            # grouping: group on the host, upload descriptor + weights
            periods, amps = sel_mod(x)
            if periods.numel() == 0:
                return unchanged()
            if periods.numel() > FTN_KMAX:
                raise ValueError(f"{periods.numel()} period candidates exceed FTN_KMAX={FTN_KMAX}")
            grp = PeriodGrouper(periods.detach().to("cpu", torch.long), amps.detach().float().cpu(), seq_len=L,
                                min_period=getattr(sel_mod, "min_period_threshold", None),
                                max_period=getattr(sel_mod, "pmax", None), block_index=self.block_index,
                                freq_indices=getattr(sel_mod, "last_frequency_indices", None)).group()
            self._lazy_sel = None
            self._last_raw_period_count = int(periods.numel())
            self._last_valid_period_count = int(grp.valid_mask.sum().item())
            self._last_group_count = int(grp.periods.numel())
            if grp.periods.numel() == 0:
                return unchanged()
            # softmax in fp32, rounded to the amplitudes' dtype and scatter-added in it, as the reference (:1000-1009)
            w = _group_weights(amps.detach().cpu(), grp.mapping, grp.periods.numel(), B).float()
            dh = lib.desc_from_periods(grp.periods.tolist(), L, 1, 2 ** 30)
            if int(dh.n_groups) != grp.periods.numel():
                raise RuntimeError("host grouping and descriptor disagree")
            sel = runtime.selection_from_host(dh, w, x.device)
        y = runtime.timesblock_forward(xf, plan, wblob, sel, norm, adt, range_flag)
        if y.dtype != x.dtype:
            y = y.to(x.dtype)
        if slot_ev is not None:
            slot_ev.record()
            self._range_pending.append((range_flag, slot_ev, x, y, post_norm))
        elif not guarded:
            self._last_engine_used = engine
        if _env_on("TIMESBLOCK_VEC_DISABLE"):
            # same kernels either way (the two reference paths are the same math, :866-953);
            # only the counters differ.  Reading the group count synchronises, as the reference does.
            self._last_loop_iterations = int(self._last_group_count)
        else:
            self._vec_calls += 1
            self._last_loop_iterations = 0
        return y

    # ---- torch backend (generic: any inception module, autograd, CPU) ----------
    def _forward_torch(self, x: torch.Tensor) -> torch.Tensor:
        periods, amps = self.period_selector(x)
        self._lazy_sel = None
        if periods.numel() == 0:
            return x
        B, L, C = x.shape
        sel_mod = self.period_selector
        amps = amps.to(device=x.device, dtype=x.dtype)
        grp = PeriodGrouper(periods.to(device=x.device, dtype=torch.long).view(-1), amps, seq_len=L,
                            min_period=getattr(sel_mod, "min_period_threshold", None),
                            max_period=getattr(sel_mod, "pmax", None), block_index=self.block_index,
                            freq_indices=getattr(sel_mod, "last_frequency_indices", None)).group()
        self._last_raw_period_count = int(periods.numel())
        self._last_valid_period_count = int(grp.valid_mask.sum().item())
        G = int(grp.periods.numel())
        self._last_group_count = G
        self._last_loop_iterations = 0
        if G == 0:
            return x
        vec = not _env_on("TIMESBLOCK_VEC_DISABLE")
        if vec:
            self._vec_calls += 1
            w = _group_weights(amps, grp.mapping, G, B)                 # softmax + scatter (:992-1009)
        else:
            w = F.softmax(grp.logits.float(), dim=1).to(x.dtype)        # loop path (:820-864)
            self._last_loop_iterations = G
        xt = x.permute(0, 2, 1)
        deltas = []
        for g in range(G):
            p, pad, cyc = int(grp.periods[g]), int(grp.pad_lengths[g]), int(grp.cycles[g])
            grid = F.pad(xt, (0, pad)).reshape(B, C, cyc, p)
            gin = grid.float() if grid.dtype != torch.float32 else grid
            out = self.inception(gin)
            deltas.append((out.float() - gin).reshape(B, C, L + pad)[..., :L].permute(0, 2, 1).to(x.dtype))
        # stacked weighted sum in the input dtype, exactly as the reference forms it (:1075-1092)
        stacked = torch.stack(deltas, dim=-1)
        combined = (stacked * w.to(dtype=stacked.dtype).view(B, 1, 1, -1)).sum(dim=-1)
        return x + combined


def _affine_layernorm(m, C: int) -> bool:
    return (isinstance(m, nn.LayerNorm) and tuple(m.normalized_shape) == (C,) and m.weight is not None
            and m.bias is not None)


def _layer_norm_fp32(module: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """LayerNorm with fp32 statistics and fp32 parameters for half inputs (reference :30-34)."""
    if isinstance(module, nn.LayerNorm):
        cd = torch.float32 if x.dtype in (torch.float16, torch.bfloat16) else x.dtype
        w = None if module.weight is None else module.weight.to(cd)
        b = None if module.bias is None else module.bias.to(cd)
        return F.layer_norm(x.to(cd), module.normalized_shape, weight=w, bias=b, eps=module.eps).to(x.dtype)
    return module(x)


def _group_weights(amps: torch.Tensor, mapping: torch.Tensor, G: int, B: int) -> torch.Tensor:
    """fp32 softmax over the valid candidates, scatter-added into their groups."""
    if amps.dim() == 1:
        amps = amps.view(1, -1)
    if amps.size(0) == 1 and B > 1:
        amps = amps.expand(B, -1)
    valid = mapping >= 0
    sm = F.softmax(amps[:, valid].float(), dim=1).to(amps.dtype)
    w = torch.zeros(sm.size(0), G, dtype=sm.dtype, device=sm.device)
    w.scatter_add_(1, mapping[valid].to(sm.device).view(1, -1).expand(sm.size(0), -1), sm)
    return w


# =========================================================================
# LowRankTemporalContext                              reference :1328-1371
# =========================================================================
class LowRankTemporalContext(nn.Module):
    """``coeff[B, N, R] -> scale * (DCT basis[L, R] @ coeff) with the time mean removed``."""

    def __init__(self, rank: int, init_scale: float = 1e-2) -> None:
        super().__init__()
        if rank <= 0:
            raise ValueError("LowRankTemporalContext requires a positive rank")
        self.rank = int(rank)
        self.scale = nn.Parameter(torch.as_tensor(float(init_scale), dtype=torch.float32))
        self.register_buffer("_cached_basis", torch.empty(0), persistent=False)
        self._cached_length = 0
        self._last_backend: Optional[str] = None

    def _compute_basis(self, length: int, device: torch.device, dtype: torch.dtype) -> torch.Tensor:
        cd = _safe_param_dtype(dtype)
        t = torch.arange(length, device=device, dtype=cd).unsqueeze(1)
        r = torch.arange(1, self.rank + 1, device=device, dtype=cd).unsqueeze(0)
        b = torch.cos(math.pi / float(length) * (t + 0.5) * r)
        b = b - b.mean(dim=0, keepdim=True)
        b = b / torch.linalg.norm(b, dim=0, keepdim=True).clamp_min(torch.finfo(b.dtype).eps)
        return b.to(dtype)

    def _basis(self, length: int, reference: torch.Tensor) -> torch.Tensor:
        if self._cached_basis.numel() == 0 or self._cached_length != length:
            self._cached_basis = self._compute_basis(length, reference.device, reference.dtype).detach()
            self._cached_length = length
        return self._cached_basis.to(device=reference.device, dtype=reference.dtype)

    def forward(self, coeff: torch.Tensor, length: int, add_to: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``add_to`` (optional, ``[B, length, N]``) fuses the caller's ``x + ctx``
        (reference :1981-1983) into the same HBM pass."""
        if coeff.ndim != 3:
            raise ValueError("LowRankTemporalContext expects coeff shaped [B, N, R]")
        if coeff.size(-1) != self.rank:
            raise ValueError("Coefficient dimension mismatch with configured rank")
        hip = (coeff.is_cuda and not (torch.is_grad_enabled() and (coeff.requires_grad or self.scale.requires_grad))
               and self.rank <= 32)
        if hip:
            from .. import runtime

            self._last_backend = "hip"
            cf = coeff.detach()
            if cf.dtype != torch.float32 or not cf.is_contiguous():
                cf = cf.float().contiguous()
            xa = None
            if add_to is not None:
                xa = add_to.detach()
                if xa.dtype != torch.float32 or not xa.is_contiguous():
                    xa = xa.float().contiguous()
            sc = self.scale.detach()
            if sc.dtype != torch.float32 or sc.device != cf.device:
                sc = sc.to(device=cf.device, dtype=torch.float32)
            out = runtime.lrtc_forward(cf, int(length), sc, xa)
            return out if out.dtype == coeff.dtype else out.to(coeff.dtype)
        self._last_backend = "torch"
        ctx = torch.einsum("lr,bnr->bln", self._basis(length, coeff), coeff)
        ctx = (ctx - ctx.mean(dim=1, keepdim=True)) * self.scale.to(device=coeff.device, dtype=coeff.dtype)
        return ctx if add_to is None else add_to + ctx


def _lrtc_project(self, coeff: torch.Tensor, length: int, weight: torch.Tensor) -> torch.Tensor:
    """``forward(coeff, length) @ weight.T`` without forming the [B, length, N] context: the map is
    linear, so ``scale * (basis - mean_l basis) @ (coeff^T weight^T)`` -> [B, length, D] (the
    value-embedding kernel adds it to ``x @ weight.T``)."""
    basis = self._basis(length, coeff)
    basis = basis - basis.mean(dim=0, keepdim=True)                    # the forward's second mean removal
    q = torch.einsum("bnr,dn->brd", coeff, weight)                      # [B, R, D]
    self._last_backend = "fused"
    return torch.matmul(basis, q) * self.scale.to(device=coeff.device, dtype=coeff.dtype)


LowRankTemporalContext.project = _lrtc_project


# The model shell (TimesNet, DataEmbedding, PositionalEmbedding, RMSNorm) lives in shell.py; it is
# reachable from here as well so that ``from ...models.timesnet import TimesNet`` keeps working.
_SHELL_NAMES = ("TimesNet", "DataEmbedding", "PositionalEmbedding", "RMSNorm")


def __getattr__(name):
    if name in _SHELL_NAMES:
        from . import shell

        return getattr(shell, name)
    raise AttributeError(name)
