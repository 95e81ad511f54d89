"""Model shell around the TimesBlock hot path: mirrors of the reference's
``PositionalEmbedding``, ``RMSNorm``, ``DataEmbedding`` and ``TimesNet``
(``models/timesnet.py:1104-1325, 1374-2102`` of the reference) with the same
constructor signatures, lazily built sub-modules, ``state_dict`` keys and
``forward(x[B,T,N], x_mark, series_static, series_ids) -> (rate, dispersion)``.

This is SURVEY §8f rank 1 ("next"): host code on PyTorch-ROCm, per the north star.
The shell itself is ordinary torch ops (Linear / LayerNorm / Embedding / softplus);
what changes is that its ``TimesBlock`` layers and its ``LowRankTemporalContext``
run on the HIP kernels of this package when the input lives on a ROCm device.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn
from torch.utils.checkpoint import checkpoint

from .timesnet import FFTPeriodSelector, LowRankTemporalContext, TimesBlock

_HALF = (torch.float16, torch.bfloat16)


def _fp32_if_half(dtype: torch.dtype) -> torch.dtype:
    return torch.float32 if dtype in _HALF else dtype


def _place(module: nn.Module, ref: torch.Tensor) -> nn.Module:
    """Lazily built modules follow the input's device; parameters stay fp32 for
    half-precision inputs (reference :30-34)."""
    return module.to(device=ref.device, dtype=_fp32_if_half(ref.dtype))


def _zeroed(linear: nn.Linear) -> nn.Linear:
    with torch.no_grad():
        linear.weight.zero_()
        if linear.bias is not None:
            linear.bias.zero_()
    return linear


# -------------------------------------------------------------------------
# embedding pieces                                       reference :1104-1325
# -------------------------------------------------------------------------
class PositionalEmbedding(nn.Module):
    """Parameter-free sinusoidal encoding, evaluated in fp32 on every call."""

    def __init__(self, d_model: int) -> None:
        super().__init__()
        self.d_model = int(d_model)
        self._table = None                     # (key, [L, d_model] fp32): pure function of (L, device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim != 3:
            raise ValueError("PositionalEmbedding expects input shaped [B, L, C]")
        B, L, _ = x.shape
        key = (L, str(x.device))
        if self._table is not None and self._table[0] == key and not torch.is_grad_enabled():
            return self._table[1].to(x.dtype).unsqueeze(0).expand(B, -1, -1)
        pos = torch.arange(L, device=x.device, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, self.d_model, 2, device=x.device, dtype=torch.float32)
                         * (-math.log(10000.0) / self.d_model))
        pe = torch.zeros(L, self.d_model, device=x.device, dtype=torch.float32)
        pe[:, 0::2] = torch.sin(pos * freq)
        n_odd = pe[:, 1::2].shape[1]
        pe[:, 1::2] = torch.cos(pos * freq[:n_odd])
        self._table = (key, pe)
        return pe.to(x.dtype).unsqueeze(0).expand(B, -1, -1)


class RMSNorm(nn.Module):
    def __init__(self, d_model: int, eps: float = 1e-5) -> None:
        super().__init__()
        if d_model <= 0:
            raise ValueError("RMSNorm expects a positive embedding dimension")
        self.eps = float(eps)
        self.weight = nn.Parameter(torch.ones(int(d_model)))
        self.bias = nn.Parameter(torch.zeros(int(d_model)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.size(-1) != self.weight.numel():
            raise ValueError("RMSNorm dimension mismatch")
        cd = _fp32_if_half(x.dtype)
        xc = x.to(cd)
        y = xc * torch.rsqrt(xc.pow(2).mean(dim=-1, keepdim=True) + self.eps)
        return (y * self.weight.to(cd) + self.bias.to(cd)).to(x.dtype)


def _norm(module: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """LayerNorm with fp32 statistics for half inputs; other modules as they are."""
    if isinstance(module, nn.LayerNorm):
        cd = _fp32_if_half(x.dtype)
        w = None if module.weight is None else module.weight.to(cd)
        b = None if module.bias is None else module.bias.to(cd)
        return F.layer_norm(x.to(cd), module.normalized_shape, weight=w, bias=b, eps=module.eps).to(x.dtype)
    return module(x)


class DataEmbedding(nn.Module):
    """value Linear(N -> d_model) + positional (+ optional time-feature Linear), with the
    reference's four normalisation modes ("decoupled" = value + gate * LayerNorm(aux))."""

    _VALID_NORM_MODES = {"none", "layer", "rms", "decoupled"}

    def __init__(self, c_in: int, d_model: int, dropout: float, time_features: Optional[int] = None,
                 use_norm: bool = True, embed_norm_mode: Optional[str] = None) -> None:
        super().__init__()
        d_model = int(d_model)
        self.value_embedding = nn.Linear(int(c_in), d_model)
        self.position_embedding = PositionalEmbedding(d_model)
        self.temporal_embedding: Optional[nn.Module] = (
            nn.Linear(int(time_features), d_model) if time_features is not None and time_features > 0 else None)
        mode = (embed_norm_mode if embed_norm_mode is not None else ("decoupled" if use_norm else "none")).lower()
        if mode not in self._VALID_NORM_MODES:
            raise ValueError(
                f"embed_norm_mode must be one of {sorted(self._VALID_NORM_MODES)}, got {embed_norm_mode!r}")
        self.embed_norm_mode = mode
        self.use_norm = mode != "none"
        self.norm: Optional[nn.Module] = None
        self.aux_norm: Optional[nn.Module] = None
        if mode == "decoupled":
            self.aux_norm = nn.LayerNorm(d_model)
            self.gate = nn.Parameter(torch.full((1, 1, d_model), 0.1, dtype=torch.float32))
        else:
            if mode == "layer":
                self.norm = nn.LayerNorm(d_model)
            elif mode == "rms":
                self.norm = RMSNorm(d_model)
            self.register_parameter("gate", None)
        self.dropout = nn.Dropout(float(dropout))

    def forward(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.ndim not in (3, 4):
            raise ValueError("DataEmbedding expects input shaped [B, L, C] or [B, L, N, C]")
        four_d = x.ndim == 4
        mark = x_mark
        if four_d:
            B, L, N, C = x.shape
            x = x.reshape(B * N, L, C)
            if mark is not None:
                if mark.ndim == 3:
                    if mark.shape[0] != B or mark.shape[1] != L:
                        raise ValueError("x_mark must match batch/time dimensions of x")
                    mark = mark.unsqueeze(2).expand(-1, -1, N, -1)
                elif mark.ndim == 4:
                    if mark.shape[:3] != (B, L, N):
                        raise ValueError("x_mark must align with [B, L, N] dimensions of x")
                else:
                    raise ValueError("x_mark must have shape [B, L, T] or [B, L, N, T]")
                mark = mark.reshape(B * N, L, mark.size(-1))
        elif mark is not None and mark.ndim != 3:
            raise ValueError("x_mark must share dimensions [B, L, T]")

        value = self.value_embedding(x)
        aux = self.position_embedding(x)
        if self.temporal_embedding is not None and mark is not None:
            aux = aux + self.temporal_embedding(mark)
        if self.embed_norm_mode == "decoupled":
            out = value + self.gate.to(value.dtype) * _norm(self.aux_norm, aux)
        else:
            out = value + aux
            if self.norm is not None:
                out = _norm(self.norm, out)
        out = self.dropout(out)
        return out.view(B, L, N, out.size(-1)) if four_d else out


# -------------------------------------------------------------------------
# TimesNet                                               reference :1374-2102
# -------------------------------------------------------------------------
class TimesNet(nn.Module):
    """Embedding -> n_layers x (TimesBlock, residual, shared LayerNorm) -> time projection
    L -> H -> rate / dispersion heads (negative-binomial parameters)."""

    def __init__(self, input_len: int, pred_len: int, d_model: int, n_layers: int, k_periods: int,
                 kernel_set, dropout: float, activation: str, mode: str, d_ff: Optional[int] = None,
                 bottleneck_ratio: float = 1.0, min_period_threshold: int = 1, channels_last: bool = False,
                 use_checkpoint: bool = True, use_embedding_norm: bool = True,
                 embed_norm_mode: Optional[str] = None, min_sigma: float = 1e-3,
                 min_sigma_vector=None, id_embed_dim: int = 32, static_proj_dim: Optional[int] = None,
                 static_layernorm: bool = True, use_zero_mean_context: bool = False, context_rank: int = 0,
                 context_scale: float = 1e-2, use_constant_context_bias: bool = False,
                 use_late_bias_head: bool = True) -> None:
        super().__init__()
        del channels_last                      # accepted for signature compatibility only
        assert mode in ("direct", "recursive")
        self.mode = mode
        self.input_len, self.pred_len = int(input_len), int(pred_len)
        self.requested_d_model = int(d_model)
        self.requested_d_ff = None if d_ff is None else int(d_ff)
        if self.requested_d_ff is not None and self.requested_d_ff <= 0:
            raise ValueError("d_ff must be a positive integer")
        self.d_model: Optional[int] = None
        self.d_ff: Optional[int] = self.requested_d_ff
        self.bottleneck_ratio = float(bottleneck_ratio)
        if self.bottleneck_ratio <= 0:
            raise ValueError("bottleneck_ratio must be a positive value")
        self.n_layers, self.dropout = int(n_layers), float(dropout)
        self.use_checkpoint = bool(use_checkpoint)
        self.use_embedding_norm = bool(use_embedding_norm)
        self.embed_norm_mode = (embed_norm_mode if embed_norm_mode is not None
                                else ("decoupled" if self.use_embedding_norm else "none"))
        self.min_sigma = float(min_sigma)
        self.k_periods = int(k_periods)
        self.kernel_set = list(kernel_set)
        self.period_selector = FFTPeriodSelector(self.k_periods, self.input_len, min_period_threshold)
        self.blocks = nn.ModuleList(
            TimesBlock(None, self.kernel_set, self.dropout, activation, d_ff=self.requested_d_ff,
                       bottleneck_ratio=self.bottleneck_ratio) for _ in range(self.n_layers))
        for i, blk in enumerate(self.blocks):
            blk.block_index = i
            object.__setattr__(blk, "period_selector", self.period_selector)   # shared, registered once
        self.residual_dropout = nn.Dropout(self.dropout)
        self.layer_norm: Optional[nn.LayerNorm] = None
        # starts as "repeat the last time step": zero weights, last input column = 1
        self.forecast_time_proj = _zeroed(nn.Linear(self.input_len, self.pred_len))
        if self.pred_len > 0:
            with torch.no_grad():
                self.forecast_time_proj.weight[:, -1] = 1.0
        self.embedding: Optional[DataEmbedding] = None
        self.embedding_time_features: Optional[int] = None
        self.mu_head: Optional[nn.Linear] = None
        self.sigma_head: Optional[nn.Linear] = None
        self.output_dim: Optional[int] = None
        self.input_channels: Optional[int] = None
        self._out_steps = self.pred_len if mode == "direct" else 1
        self.register_buffer("min_sigma_vector", None)
        if min_sigma_vector is not None:
            self.min_sigma_vector = torch.as_tensor(min_sigma_vector, dtype=torch.float32).reshape(1, 1, -1)
        self.id_embed_dim = int(id_embed_dim)
        if self.id_embed_dim < 0:
            raise ValueError("id_embed_dim must be non-negative")
        self.static_proj_dim = None if static_proj_dim is None else int(static_proj_dim)
        if self.static_proj_dim is not None and self.static_proj_dim <= 0:
            raise ValueError("static_proj_dim must be a positive integer when provided")
        self.static_layernorm = bool(static_layernorm)
        self.series_embedding: Optional[nn.Embedding] = None
        self.static_proj: Optional[nn.Linear] = None
        self.static_norm: Optional[nn.Module] = None
        self.context_norm: Optional[nn.LayerNorm] = None
        self.context_proj: Optional[nn.Linear] = None
        self.context_coeff: Optional[nn.Linear] = None
        self.temporal_context: Optional[LowRankTemporalContext] = None
        self.late_bias_norm: Optional[nn.LayerNorm] = None
        self.late_bias_head: Optional[nn.Linear] = None
        self.register_parameter("late_bias_gate", None)
        self.pre_embedding_norm: Optional[nn.Module] = None
        self.pre_embedding_dropout = nn.Dropout(self.dropout)
        self._static_in_features: Optional[int] = None
        self._static_out_dim = 0
        self._series_id_vocab: Optional[int] = None
        self._series_id_reference: Optional[torch.Tensor] = None
        self.debug_memory = False
        self.use_zero_mean_context = bool(use_zero_mean_context)
        self.use_constant_context_bias = bool(use_constant_context_bias)
        self.use_late_bias_head = bool(use_late_bias_head)
        self.context_rank = int(context_rank)
        if self.context_rank < 0:
            raise ValueError("context_rank must be non-negative")
        self.context_scale_default = float(context_scale)
        self._last_head_backend = "torch"
        self._last_embed_backend = "torch"
        self._defer_checks = False
        self._pending_bad = None

    # ---- lazy construction ------------------------------------------------------
    def _lazy(self, name: str, ref: torch.Tensor, ok: Callable[[nn.Module], bool],
              make: Callable[[], nn.Module]) -> nn.Module:
        """(Re)build attribute ``name`` when it is missing or ``ok`` rejects it; always move it
        to the input's device with fp32-safe parameters."""
        cur = getattr(self, name)
        if cur is None or not ok(cur):
            cur = make()
        cur = _place(cur, ref)
        setattr(self, name, cur)
        return cur

    def _ensure_embedding(self, x, x_mark=None, series_static=None, series_ids=None) -> None:
        n_series = int(x.size(-1))
        time_dim = int(x_mark.size(-1)) if x_mark is not None else 0
        if self.input_channels is None:
            self.input_channels = n_series
        elif self.input_channels != n_series:
            raise ValueError("Number of series changed between calls")
        if self.d_model is None:
            self.d_model = self.requested_d_model
        elif self.d_model != self.requested_d_model:
            raise ValueError("d_model changed between calls")
        self.d_ff = self.d_model if self.requested_d_ff is None else self.requested_d_ff

        # -- static covariates
        static_dim = 0
        if series_static is not None:
            if series_static.ndim not in (2, 3):
                raise ValueError("series_static must have shape [N, F] or [B, N, F]")
            ref_rows = series_static if series_static.ndim == 2 else series_static[0]
            if ref_rows.size(0) != n_series:
                raise ValueError("series_static must align with the number of input series")
            feat = int(ref_rows.size(-1))
            if feat <= 0:
                raise ValueError("series_static must have at least one feature")
            if self.static_proj is None:
                width = self.static_proj_dim if self.static_proj_dim is not None else feat
                self.static_proj = _place(nn.Linear(feat, width), x)
                self.static_norm = _place(nn.LayerNorm(width), x) if self.static_layernorm else nn.Identity()
                self._static_in_features = feat
            else:
                if self.static_proj.in_features != feat:
                    raise ValueError("series_static feature dimension changed between calls")
                self.static_proj = _place(self.static_proj, x)
                if self.static_norm is not None:
                    self.static_norm = _place(self.static_norm, x)
            static_dim = int(self.static_proj.out_features)
        elif self.static_proj is not None:
            self.static_proj = _place(self.static_proj, x)
            if self.static_norm is not None:
                self.static_norm = _place(self.static_norm, x)
            static_dim = int(self.static_proj.out_features)
        self._static_out_dim = static_dim

        # -- series identifiers
        id_dim = 0
        if self.id_embed_dim > 0:
            ids_ref = None
            if series_ids is not None:
                if series_ids.ndim == 1:
                    ids_ref = series_ids.to(torch.long)
                elif series_ids.ndim == 2:
                    ids_ref = series_ids[0].to(torch.long)
                else:
                    raise ValueError("series_ids must have shape [N] or [B, N]")
                if ids_ref.numel() != n_series:
                    raise ValueError("series_ids length must match number of series")
            vocab_of = lambda t: int(t.max().item()) + 1 if t.numel() > 0 else n_series
            if self.series_embedding is None:
                if ids_ref is None:
                    ids_ref = torch.arange(n_series, device=x.device, dtype=torch.long)
                self.series_embedding = _place(nn.Embedding(vocab_of(ids_ref), self.id_embed_dim), x)
                self._series_id_reference = ids_ref.to(device=x.device)
            else:
                self.series_embedding = _place(self.series_embedding, x)
                if ids_ref is not None:
                    if not self._defer_checks and vocab_of(ids_ref) > int(self.series_embedding.num_embeddings):
                        raise ValueError("series_ids vocabulary expanded between calls")
                    self._series_id_reference = ids_ref.to(device=x.device)
                elif self._series_id_reference is None:
                    self._series_id_reference = torch.arange(n_series, device=x.device, dtype=torch.long)
            self._series_id_vocab = int(self.series_embedding.num_embeddings)
            if self._series_id_reference is not None and self._series_id_reference.numel() != n_series:
                raise ValueError("series identifier count changed between calls")
            id_dim = int(self.series_embedding.embedding_dim)

        # -- context consumers (LRTC coefficients, constant bias, late bias head)
        ctx = static_dim + id_dim
        steps = self._out_steps
        if ctx > 0:
            self._lazy("context_norm", x, lambda m: tuple(m.normalized_shape) == (ctx,), lambda: nn.LayerNorm(ctx))
            if self.use_zero_mean_context and self.context_rank > 0:
                self._lazy("context_coeff", x,
                           lambda m: m.in_features == ctx and m.out_features == self.context_rank,
                           lambda: _zeroed(nn.Linear(ctx, self.context_rank)))
                self._lazy("temporal_context", x, lambda m: m.rank == self.context_rank,
                           lambda: LowRankTemporalContext(self.context_rank, self.context_scale_default))
            else:
                self.context_coeff = None
                self.temporal_context = None
            if self.use_constant_context_bias:
                self._lazy("context_proj", x, lambda m: m.in_features == ctx, lambda: _zeroed(nn.Linear(ctx, 1)))
            else:
                self.context_proj = None
            if self.use_late_bias_head:
                self._lazy("late_bias_norm", x, lambda m: tuple(m.normalized_shape) == (ctx,),
                           lambda: nn.LayerNorm(ctx))
                self._lazy("late_bias_head", x, lambda m: m.in_features == ctx and m.out_features == steps,
                           lambda: _zeroed(nn.Linear(ctx, steps)))
                gate = self.late_bias_gate
                if not isinstance(gate, nn.Parameter) or tuple(gate.shape) != (1, steps, 1):
                    self.late_bias_gate = nn.Parameter(
                        torch.full((1, steps, 1), 0.05, dtype=torch.float32, device=x.device))
                else:
                    gate.data = gate.data.to(device=x.device, dtype=torch.float32)
            else:
                self.late_bias_norm = None
                self.late_bias_head = None
                if isinstance(self.late_bias_gate, nn.Parameter):
                    self.late_bias_gate = None
        else:
            self.context_norm = self.context_proj = self.context_coeff = None
            self.temporal_context = None
            self.late_bias_norm = self.late_bias_head = None
            if isinstance(self.late_bias_gate, nn.Parameter):
                self.late_bias_gate = None

        # -- built for checkpoint compatibility; the reference never applies it in forward
        if ctx == 0:
            if not isinstance(self.pre_embedding_norm, nn.Identity):
                self.pre_embedding_norm = nn.Identity()
            self.pre_embedding_norm = self.pre_embedding_norm.to(device=x.device)
        else:
            keep = (isinstance(self.pre_embedding_norm, nn.LayerNorm)
                    and tuple(self.pre_embedding_norm.normalized_shape) == (1 + ctx,))
            self.pre_embedding_norm = _place(self.pre_embedding_norm if keep else nn.LayerNorm(1 + ctx), x)
        self.pre_embedding_dropout = self.pre_embedding_dropout.to(device=x.device)

        msv = self.min_sigma_vector
        if isinstance(msv, torch.Tensor) and msv.numel() > 0:
            if int(msv.shape[-1]) < n_series:
                raise ValueError("min_sigma_vector length does not match number of series")
            if int(msv.shape[-1]) != n_series:
                self.min_sigma_vector = msv[..., :n_series]

        if self.embedding_time_features is not None and self.embedding_time_features != time_dim:
            raise ValueError("Temporal feature dimension changed between calls")
        self._lazy("embedding", x, lambda m: True,
                   lambda: DataEmbedding(n_series, self.d_model, self.dropout,
                                         time_features=time_dim if time_dim > 0 else None,
                                         use_norm=self.use_embedding_norm, embed_norm_mode=self.embed_norm_mode))
        self.embedding_time_features = time_dim
        self.forecast_time_proj = _place(self.forecast_time_proj, x)
        self._lazy("layer_norm", x, lambda m: tuple(m.normalized_shape) == (self.d_model,),
                   lambda: nn.LayerNorm(self.d_model))
        head_ok = lambda m: m.in_features == self.d_model and m.out_features == n_series
        # zero heads: the first forward reproduces "softplus(last observed values)"
        self._lazy("mu_head", x, head_ok, lambda: _zeroed(nn.Linear(self.d_model, n_series)))
        self._lazy("sigma_head", x, head_ok, lambda: _zeroed(nn.Linear(self.d_model, n_series)))
        self.output_dim = self.input_channels

    def _dispersion_floor_from_ref(self, ref: torch.Tensor) -> torch.Tensor:
        msv = self.min_sigma_vector
        if isinstance(msv, torch.Tensor) and msv.numel() > 0:
            return msv.to(device=ref.device, dtype=ref.dtype).expand_as(ref).clone()
        return ref.new_full(ref.shape, self.min_sigma)

    # ---- HIP value embedding (ftn_embed_forward) --------------------------------------
    def _hip_embed_ok(self, window: torch.Tensor) -> bool:
        emb = self.embedding
        if emb is None or emb.embed_norm_mode not in ("decoupled", "none", "layer"):
            return False
        if emb.embed_norm_mode == "layer" and not (isinstance(emb.norm, nn.LayerNorm) and emb.norm.weight is not None
                                                   and emb.norm.bias is not None):
            return False
        return (window.is_cuda and window.dtype == torch.float32
                and not (torch.is_grad_enabled() and (window.requires_grad or emb.value_embedding.weight.requires_grad))
                and not (self.training and self.dropout > 0.0)
                and self.d_model % 4 == 0 and self.d_model <= 128
                and window.stride(2) == 1 and window.stride(1) == window.size(2))

    def _embed_hip(self, window, mark, coeff, cb) -> torch.Tensor:
        """DataEmbedding.forward with the context front-end folded in: one pass over the window,
        ``x W^T + add`` where ``add`` carries bias + positional / time-feature term + the temporal
        context and constant bias pushed through W (reference :1958-1996, :1283-1325)."""
        from .. import runtime

        emb = self.embedding
        L = window.size(1)
        wv = emb.value_embedding.weight.detach()
        aux = emb.position_embedding(window[:1])                             # [1, L, d]
        if emb.temporal_embedding is not None and mark is not None:
            aux = aux + emb.temporal_embedding(mark)                         # [B, L, d]
        if emb.embed_norm_mode == "decoupled":
            aux = emb.gate.to(aux.dtype) * _norm(emb.aux_norm, aux)
        add = aux + emb.value_embedding.bias.detach()
        if coeff is not None:
            add = add + self.temporal_context.project(coeff.detach().float(), L, wv)
        if cb is not None:
            add = add + torch.matmul(cb.detach().float(), wv.t()).unsqueeze(1)
        norm = None
        if emb.embed_norm_mode == "layer":
            norm = (emb.norm.weight.detach().float().contiguous(), emb.norm.bias.detach().float().contiguous(),
                    emb.norm.eps)
        self._last_embed_backend = "hip"
        return runtime.embed_forward(window, wv, add.detach().float().contiguous(), norm)

    # ---- HIP heads (ftn_head_forward) ----------------------------------------------
    def _hip_heads_ok(self, seq: torch.Tensor, window: torch.Tensor) -> bool:
        params = (self.forecast_time_proj.weight, self.mu_head.weight, self.sigma_head.weight)
        return (seq.is_cuda and seq.dtype == torch.float32 and window.dtype == torch.float32
                and not (torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in params)))
                and self.d_model % 4 == 0 and self.d_model <= 128
                and window.stride(2) == 1 and window.stride(1) == window.size(2))

    def _heads_hip(self, seq, window, late, steps: int, hist: int):
        """Time projection as one batched GEMM straight into [B, steps, d_model] (no permute copies),
        then mu / sigma heads + history tail + late bias + softplus + floors + the finite-positive
        check in one kernel (reference :2066-2102)."""
        from .. import runtime

        B, L, _ = seq.shape
        wt, bt = self.forecast_time_proj.weight, self.forecast_time_proj.bias
        if steps != self.pred_len:
            wt, bt = wt[-steps:], bt[-steps:]
        hidden = torch.baddbmm(bt.detach().view(1, -1, 1), wt.detach().unsqueeze(0).expand(B, -1, -1), seq.detach())
        msv = self.min_sigma_vector
        floor_vec = None
        if isinstance(msv, torch.Tensor) and msv.numel() > 0:
            floor_vec = msv.to(device=seq.device, dtype=torch.float32).reshape(-1).contiguous()
        late_c = None if late is None else late.detach().float().contiguous()
        rate, dispersion, bad = runtime.head_forward(
            hidden.contiguous(), self.mu_head.weight.detach(), self.mu_head.bias.detach(),
            self.sigma_head.weight.detach(), self.sigma_head.bias.detach(), window[:, -hist:, :], hist, late_c,
            floor_vec, self.min_sigma)
        self._last_head_backend = "hip"
        self._pending_bad = bad
        if not self._defer_checks:
            self.check_outputs()
        return rate, dispersion

    def check_outputs(self) -> None:
        """Raise the reference's RuntimeError (:2095-2098) if the last HIP head call produced a rate or
        dispersion that is not finite and > 0.  Reads one int32 from the device (synchronises); called by
        ``forward`` itself unless ``_defer_checks`` is set (HIP-graph capture, ``graph.GraphedForward``)."""
        # f16x2 range guard of the blocks (TimesBlock.check_range): a block that had to repeat its call on bf16x3 has
        # repaired its own output, but the layers behind it have consumed the old one - forward() then runs again
        # with every block on bf16x3 (which stays selected)
        seen = sum(getattr(blk, "_range_fallbacks", 0) for blk in self.blocks)
        for blk in self.blocks:
            if getattr(blk, "_range_dev_flag", None) is not None or getattr(blk, "_range_pending", None):
                blk.check_range()
        if sum(getattr(blk, "_range_fallbacks", 0) for blk in self.blocks) != seen:
            for blk in self.blocks:
                blk.engine = "bf16x3"
            self._range_retry = True
            self._pending_bad = None
            return
        bad, self._pending_bad = self._pending_bad, None
        if bad is None:
            return
        flag = int(bad.item())
        if flag & 1:
            raise RuntimeError("Predicted rate must be finite and strictly positive")
        if flag & 2:
            raise RuntimeError("Predicted dispersion must be finite and strictly positive")

    # ---- forward ------------------------------------------------------------------
    def forward(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None,
                series_static: Optional[torch.Tensor] = None,
                series_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        out = self._forward_once(x, x_mark, series_static, series_ids)
        if getattr(self, "_range_retry", False):                # check_outputs(): a block left the f16x2 range
            self._range_retry = False
            out = self._forward_once(x, x_mark, series_static, series_ids)
        return out

    def _forward_once(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None,
                      series_static: Optional[torch.Tensor] = None,
                      series_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if x.ndim != 3:
            raise ValueError("TimesNet expects input shaped [B, T, N]")
        B, T, N = x.shape
        if T < self.input_len:
            raise ValueError(f"Input sequence length {T} is shorter than required input_len {self.input_len}")
        mark = None
        if x_mark is not None:
            if x_mark.shape[:2] != x.shape[:2]:
                raise ValueError("x_mark must share batch/time dimensions with x")
            mark = x_mark[:, -self.input_len:, :]
        window = x[:, -self.input_len:, :]
        self._ensure_embedding(window, mark, series_static, series_ids)
        steps = self.pred_len if self.mode == "direct" else self._out_steps
        L = window.size(1)
        feats_in = window

        # -- per-series context vector [Bc, N, ctx].  Every op on it is row-wise, so when the static
        #    features / ids are shared by the batch (2-D / 1-D inputs, the pipeline's case) it is built
        #    once (Bc = 1) and broadcast; the reference expands to B first (:1883-1956), same values.
        shared = ((series_static is None or series_static.ndim == 2)
                  and (series_ids is None or series_ids.ndim == 1 or series_ids.size(0) == 1))
        Bc = 1 if shared else B
        parts = []
        if self.static_proj is not None and series_static is not None:
            if series_static.ndim == 2:
                st = series_static.unsqueeze(0).expand(Bc, -1, -1)
            elif series_static.ndim == 3:
                if series_static.size(0) != B:
                    raise ValueError("series_static batch dimension must match input batch size")
                st = series_static
            else:
                raise ValueError("series_static must have shape [N, F] or [B, N, F]")
            sp = self.static_proj(st.to(device=window.device, dtype=window.dtype, non_blocking=window.is_cuda))
            parts.append(_norm(self.static_norm, sp) if self.static_norm is not None else sp)
        if self.series_embedding is not None and self.id_embed_dim > 0:
            if series_ids is None:
                if self._series_id_reference is None:
                    ids = torch.arange(N, device=window.device, dtype=torch.long).unsqueeze(0)
                else:
                    ids = self._series_id_reference.view(1, -1).to(window.device)
                    if ids.size(1) != N:
                        raise ValueError("Stored series identifiers do not match input dimension")
            else:
                ids = series_ids.unsqueeze(0) if series_ids.ndim == 1 else series_ids
                if ids.ndim != 2:
                    raise ValueError("series_ids must have shape [N] or [B, N]")
            if ids.size(0) == 1 and Bc > 1:
                ids = ids.expand(Bc, -1)
            if series_ids is not None:
                if ids.size(0) not in (1, B):
                    raise ValueError("series_ids batch dimension does not match input")
                if ids.size(1) != N:
                    raise ValueError("series_ids length must match number of series")
                ids = ids.to(device=window.device, dtype=torch.long)
                self._series_id_reference = ids[0].detach().clone()
            parts.append(self.series_embedding(ids.to(device=window.device, dtype=torch.long)))

        ctx = coeff = cb = None
        fused_embed = self._hip_embed_ok(window)
        if parts:
            ctx = torch.cat(parts, dim=-1)
            if self.context_norm is not None:
                ctx = _norm(self.context_norm, ctx)
            if self.use_zero_mean_context and self.context_coeff is not None and self.temporal_context is not None:
                coeff = self.context_coeff(ctx.to(self.context_coeff.weight.dtype))
                if not fused_embed:
                    signal = self.temporal_context(coeff, L)          # HIP LRTC kernel on ROCm tensors
                    if signal.ndim != 3 or signal.shape != (Bc,) + tuple(feats_in.shape[1:]):
                        raise RuntimeError("Temporal context must align with the [B, L, N] input")
                    feats_in = feats_in + signal.to(feats_in.dtype)
            if self.use_constant_context_bias and self.context_proj is not None:
                cb = self.context_proj(ctx.to(self.context_proj.weight.dtype)).squeeze(-1)
                if not fused_embed:
                    feats_in = feats_in + cb.to(feats_in.dtype).unsqueeze(1)

        if fused_embed:
            seq = self._embed_hip(window, mark, coeff, cb)
        else:
            seq = self.embedding(feats_in, mark)
        if seq.ndim != 3 or seq.size(1) != self.input_len or seq.size(-1) != self.d_model:
            raise RuntimeError("Embedding output must have shape [B, input_len, d_model]")

        hist = min(steps, L)
        tail = window[:, -hist:, :]
        if hist < steps:
            tail = torch.cat([tail, tail[:, -1:, :].expand(-1, steps - hist, -1)], dim=1)

        self.period_selector = self.period_selector.to(device=seq.device, dtype=seq.dtype)
        for blk in self.blocks:
            object.__setattr__(blk, "period_selector", self.period_selector)
        recompute = self.use_checkpoint and torch.is_grad_enabled()
        fuse_norm = not recompute and not (self.training and self.dropout > 0.0)
        for blk in self.blocks:
            if fuse_norm:
                # eval: dropout is the identity, so residual + shared LayerNorm ride in the block's last kernel
                seq = blk(seq, post_norm=self.layer_norm)
                continue
            new = checkpoint(blk, seq, use_reentrant=False) if recompute else blk(seq)
            seq = _norm(self.layer_norm, seq + self.residual_dropout(new - seq))

        late = None
        if (ctx is not None and self.late_bias_head is not None and self.late_bias_norm is not None
                and isinstance(self.late_bias_gate, nn.Parameter)):
            c = ctx.to(dtype=self.late_bias_head.weight.dtype, device=self.late_bias_head.weight.device)
            lb = self.late_bias_head(_norm(self.late_bias_norm, c)).permute(0, 2, 1)      # [Bc, steps, N]
            late = self.late_bias_gate.to(dtype=lb.dtype, device=lb.device) * lb

        if self._hip_heads_ok(seq, window):
            return self._heads_hip(seq, window, late, steps, hist)

        # -- time projection L -> pred_len on [B, d_model, L], heads back on [B, steps, d_model]
        proj = self.forecast_time_proj(seq.permute(0, 2, 1).contiguous())
        if steps != self.pred_len:
            proj = proj[:, :, -steps:]
        hidden = proj.permute(0, 2, 1).contiguous()
        pre = self.mu_head(hidden) + tail.to(window.dtype)
        if late is not None:
            pre = pre + late.to(pre.dtype)
        rate = F.softplus(pre.float(), beta=1.0, threshold=20).to(pre.dtype) + 1e-6
        sig = self.sigma_head(hidden)
        sig = F.softplus(sig.float(), beta=1.0, threshold=20).to(sig.dtype)
        dispersion = sig + self._dispersion_floor_from_ref(rate).to(sig.dtype) + 1e-6
        if torch.any(~torch.isfinite(rate)) or torch.any(rate <= 0):
            raise RuntimeError("Predicted rate must be finite and strictly positive")
        if torch.any(~torch.isfinite(dispersion)) or torch.any(dispersion <= 0):
            raise RuntimeError("Predicted dispersion must be finite and strictly positive")
        if rate.shape != (B, steps, N) or dispersion.shape != (B, steps, N):
            raise RuntimeError("Predicted rate/dispersion have incorrect shape")
        return rate, dispersion
