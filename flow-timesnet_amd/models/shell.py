"""Model shell around the TimesBlock hot path: mirrors of the reference's ``PositionalEmbedding``, ``RMSNorm``,
``DataEmbedding`` and ``TimesNet`` (``models/timesnet.py:1104-1325, 1374-2102`` of the reference) - the same
constructor signatures, attribute names, error messages and ``state_dict`` keys (the checkpoint ABI), and
``forward(x[B,T,N], x_mark, series_static, series_ids) -> (rate, dispersion)``.

SURVEY section 8f ranks 1-2.  The structure is this package's own: the lazily built layers come from one declarative
table (``_LAZY_TABLE``), and the forward is four stages - series context rows, embedding front end, block stack,
heads - each of which calls the HIP entry point for its stage when the tensors live on a ROCm device
(``ftn_embed_forward``, the TimesBlock kernels with the LayerNorm epilogue, ``ftn_head_forward``) and composes
stock torch ops otherwise (CPU tensors, autograd, dropout in training).
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple, Optional, Tuple

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


def _norm(module: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """LayerNorm with fp32 statistics for half inputs; other modules as they are."""
    if not isinstance(module, nn.LayerNorm):
        return module(x)
    cd = _fp32_if_half(x.dtype)
    cast = lambda p: None if p is None else p.to(cd)
    return F.layer_norm(x.to(cd), module.normalized_shape, cast(module.weight), cast(module.bias), module.eps).to(x.dtype)


# -------------------------------------------------------------------------
# embedding pieces                                       reference :1104-1325
# -------------------------------------------------------------------------
class PositionalEmbedding(nn.Module):
    """Parameter-free sinusoidal encoding ``[sin(t w_0), cos(t w_0), sin(t w_1), ...]``, ``w_i = 10000^(-2i/d)``,
    evaluated in fp32; the table is a pure function of (L, device) and is kept between calls."""

    def __init__(self, d_model: int) -> None:
        super().__init__()
        self.d_model = int(d_model)
        self._table: Optional[Tuple[Tuple[int, str], torch.Tensor]] = None

    def table(self, L: int, device: torch.device) -> torch.Tensor:
        key = (L, str(device))
        if self._table is None or self._table[0] != key or torch.is_grad_enabled():
            d = self.d_model
            rate = torch.exp(torch.arange(0, d, 2, device=device, dtype=torch.float32) * (-math.log(10000.0) / d))
            phase = torch.arange(L, device=device, dtype=torch.float32).unsqueeze(1) * rate          # [L, ceil(d/2)]
            pe = torch.stack((torch.sin(phase), torch.cos(phase)), dim=-1).reshape(L, -1)[:, :d].contiguous()
            self._table = (key, pe)
        return self._table[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim != 3:
            raise ValueError("PositionalEmbedding expects input shaped [B, L, C]")
        return self.table(x.size(1), x.device).to(x.dtype).unsqueeze(0).expand(x.size(0), -1, -1)


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
        v = x.to(cd)
        inv_rms = torch.rsqrt(v.square().mean(dim=-1, keepdim=True) + self.eps)
        return torch.addcmul(self.bias.to(cd), v * inv_rms, self.weight.to(cd)).to(x.dtype)


class DataEmbedding(nn.Module):
    """value Linear(N -> d_model) + positional (+ optional time-feature Linear), with the
    reference's four normalisation modes ("decoupled" = value + gate * LayerNorm(aux))."""

    _VALID_NORM_MODES = {"none", "layer", "rms", "decoupled"}

    def __init__(self, c_in: int, d_model: int, dropout: float, time_features: Optional[int] = None,
                 use_norm: bool = True, embed_norm_mode: Optional[str] = None) -> None:
        super().__init__()
        d_model = int(d_model)
        mode = (embed_norm_mode if embed_norm_mode is not None else ("decoupled" if use_norm else "none")).lower()
        if mode not in self._VALID_NORM_MODES:
            raise ValueError(
                f"embed_norm_mode must be one of {sorted(self._VALID_NORM_MODES)}, got {embed_norm_mode!r}")
        self.embed_norm_mode = mode
        self.use_norm = mode != "none"
        self.value_embedding = nn.Linear(int(c_in), d_model)
        self.position_embedding = PositionalEmbedding(d_model)
        self.temporal_embedding: Optional[nn.Module] = (
            nn.Linear(int(time_features), d_model) if time_features is not None and time_features > 0 else None)
        # exactly one of (aux_norm + gate) / norm / nothing, by mode; `gate` is always a registered name
        self.norm: Optional[nn.Module] = {"layer": lambda: nn.LayerNorm(d_model), "rms": lambda: RMSNorm(d_model)}.get(
            mode, lambda: None)()
        self.aux_norm: Optional[nn.Module] = nn.LayerNorm(d_model) if mode == "decoupled" else None
        if mode == "decoupled":
            self.gate = nn.Parameter(torch.full((1, 1, d_model), 0.1, dtype=torch.float32))
        else:
            self.register_parameter("gate", None)
        self.dropout = nn.Dropout(float(dropout))

    @staticmethod
    def _fold_series(x: torch.Tensor, mark: Optional[torch.Tensor]):
        """[B, L, N, C] input: every series becomes its own sample (the reference's plain reshape, :1290-1309)."""
        B, L, N, C = x.shape
        if mark is not None:
            if mark.ndim == 3:
                if tuple(mark.shape[:2]) != (B, L):
                    raise ValueError("x_mark must match batch/time dimensions of x")
                mark = mark.unsqueeze(2).expand(-1, -1, N, -1)
            elif mark.ndim != 4:
                raise ValueError("x_mark must have shape [B, L, T] or [B, L, N, T]")
            elif tuple(mark.shape[:3]) != (B, L, N):
                raise ValueError("x_mark must align with [B, L, N] dimensions of x")
            mark = mark.reshape(B * N, L, mark.size(-1))
        return x.reshape(B * N, L, C), mark

    def aux_term(self, x: torch.Tensor, mark: Optional[torch.Tensor]) -> torch.Tensor:
        """Everything added to the value embedding: positions (+ time features), gated and normalised in the
        "decoupled" mode.  [1 or B, L, d_model]."""
        aux = self.position_embedding(x)
        if mark is not None and self.temporal_embedding is not None:
            aux = aux + self.temporal_embedding(mark)
        if self.aux_norm is not None:
            aux = self.gate.to(aux.dtype) * _norm(self.aux_norm, aux)
        return aux

    def forward(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None) -> torch.Tensor:
        lead = None
        if x.ndim == 4:
            lead = tuple(x.shape[:3])
            x, x_mark = self._fold_series(x, x_mark)
        elif x.ndim != 3:
            raise ValueError("DataEmbedding expects input shaped [B, L, C] or [B, L, N, C]")
        elif x_mark is not None and x_mark.ndim != 3:
            raise ValueError("x_mark must share dimensions [B, L, T]")
        out = self.value_embedding(x) + self.aux_term(x, x_mark)
        if self.norm is not None:
            out = _norm(self.norm, out)
        out = self.dropout(out)
        return out if lead is None else out.view(*lead, out.size(-1))


# -------------------------------------------------------------------------
# TimesNet                                               reference :1374-2102
# -------------------------------------------------------------------------
class _Dims(NamedTuple):
    """What the lazily built layers are sized from (one call's view of the inputs)."""
    n_series: int
    time_dim: int
    static_dim: int
    id_dim: int
    steps: int
    d_model: int

    @property
    def ctx(self) -> int:
        return self.static_dim + self.id_dim


def _zero_linear(i: int, o: int) -> nn.Linear:
    lin = nn.Linear(i, o)
    with torch.no_grad():
        lin.weight.zero_()
        lin.bias.zero_()
    return lin


def _ln_of(width: int) -> Callable[[nn.Module], bool]:
    return lambda m: isinstance(m, nn.LayerNorm) and tuple(m.normalized_shape) == (width,)


def _lin_of(i: int, o: int) -> Callable[[nn.Module], bool]:
    return lambda m: isinstance(m, nn.Linear) and (m.in_features, m.out_features) == (i, o)


class _Lazy(NamedTuple):
    """One lazily built sub-module of TimesNet: attribute name, whether this configuration has it at all, whether an
    existing instance still fits the current sizes, and how to build it.  The state_dict keys (= the attribute names)
    and the initial values (zero heads / zero context maps, reference :1660-1662, :1719-1720, :1825-1842) are the
    checkpoint contract; the order of the table is the order of first construction."""
    name: str
    wanted: Callable[["TimesNet", _Dims], bool]
    fits: Callable[["TimesNet", _Dims], Callable[[nn.Module], bool]]
    make: Callable[["TimesNet", _Dims], nn.Module]


_HAS_CTX = lambda net, d: d.ctx > 0
_LAZY_TABLE: Tuple[_Lazy, ...] = (
    _Lazy("context_norm", _HAS_CTX, lambda net, d: _ln_of(d.ctx), lambda net, d: nn.LayerNorm(d.ctx)),
    _Lazy("context_coeff", lambda net, d: d.ctx > 0 and net.use_zero_mean_context and net.context_rank > 0,
          lambda net, d: _lin_of(d.ctx, net.context_rank), lambda net, d: _zero_linear(d.ctx, net.context_rank)),
    _Lazy("temporal_context", lambda net, d: d.ctx > 0 and net.use_zero_mean_context and net.context_rank > 0,
          lambda net, d: (lambda m: getattr(m, "rank", None) == net.context_rank),
          lambda net, d: LowRankTemporalContext(net.context_rank, net.context_scale_default)),
    _Lazy("context_proj", lambda net, d: d.ctx > 0 and net.use_constant_context_bias,
          lambda net, d: _lin_of(d.ctx, 1), lambda net, d: _zero_linear(d.ctx, 1)),
    _Lazy("late_bias_norm", lambda net, d: d.ctx > 0 and net.use_late_bias_head,
          lambda net, d: _ln_of(d.ctx), lambda net, d: nn.LayerNorm(d.ctx)),
    _Lazy("late_bias_head", lambda net, d: d.ctx > 0 and net.use_late_bias_head,
          lambda net, d: _lin_of(d.ctx, d.steps), lambda net, d: _zero_linear(d.ctx, d.steps)),
    # built for checkpoint compatibility only: the reference never applies it in forward (:1754-1774)
    _Lazy("pre_embedding_norm", lambda net, d: True,
          lambda net, d: ((lambda m: isinstance(m, nn.Identity)) if d.ctx == 0 else _ln_of(1 + d.ctx)),
          lambda net, d: nn.Identity() if d.ctx == 0 else nn.LayerNorm(1 + d.ctx)),
    _Lazy("embedding", lambda net, d: True, lambda net, d: (lambda m: True),
          lambda net, d: DataEmbedding(d.n_series, d.d_model, net.dropout, time_features=d.time_dim or None,
                                       use_norm=net.use_embedding_norm, embed_norm_mode=net.embed_norm_mode)),
    _Lazy("layer_norm", lambda net, d: True, lambda net, d: _ln_of(d.d_model), lambda net, d: nn.LayerNorm(d.d_model)),
    _Lazy("mu_head", lambda net, d: True, lambda net, d: _lin_of(d.d_model, d.n_series),
          lambda net, d: _zero_linear(d.d_model, d.n_series)),
    _Lazy("sigma_head", lambda net, d: True, lambda net, d: _lin_of(d.d_model, d.n_series),
          lambda net, d: _zero_linear(d.d_model, d.n_series)),
)


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
        # -- validated scalars
        checks = ((d_ff is None or int(d_ff) > 0, "d_ff must be a positive integer"),
                  (float(bottleneck_ratio) > 0, "bottleneck_ratio must be a positive value"),
                  (int(id_embed_dim) >= 0, "id_embed_dim must be non-negative"),
                  (static_proj_dim is None or int(static_proj_dim) > 0,
                   "static_proj_dim must be a positive integer when provided"),
                  (int(context_rank) >= 0, "context_rank must be non-negative"))
        for ok, msg in checks:
            if not ok:
                raise ValueError(msg)
        self.mode = mode
        self.input_len, self.pred_len = int(input_len), int(pred_len)
        self.requested_d_model = int(d_model)
        self.requested_d_ff = None if d_ff is None else int(d_ff)
        self.d_model: Optional[int] = None
        self.d_ff: Optional[int] = self.requested_d_ff
        self.bottleneck_ratio = float(bottleneck_ratio)
        self.n_layers, self.dropout = int(n_layers), float(dropout)
        self.use_checkpoint = bool(use_checkpoint)
        self.use_embedding_norm = bool(use_embedding_norm)
        self.embed_norm_mode = embed_norm_mode or ("decoupled" if self.use_embedding_norm else "none")
        self.min_sigma = float(min_sigma)
        self.k_periods = int(k_periods)
        self.kernel_set = list(kernel_set)
        self.id_embed_dim = int(id_embed_dim)
        self.static_proj_dim = None if static_proj_dim is None else int(static_proj_dim)
        self.static_layernorm = bool(static_layernorm)
        self.use_zero_mean_context = bool(use_zero_mean_context)
        self.use_constant_context_bias = bool(use_constant_context_bias)
        self.use_late_bias_head = bool(use_late_bias_head)
        self.context_rank = int(context_rank)
        self.context_scale_default = float(context_scale)
        self.debug_memory = False
        self._out_steps = self.pred_len if mode == "direct" else 1

        # -- modules that exist from the start
        self.period_selector = FFTPeriodSelector(self.k_periods, self.input_len, min_period_threshold)
        self.blocks = nn.ModuleList(
            TimesBlock(None, self.kernel_set, self.dropout, activation, d_ff=self.requested_d_ff,
                       bottleneck_ratio=self.bottleneck_ratio) for _ in range(self.n_layers))
        for depth, blk in enumerate(self.blocks):
            blk.block_index = depth
            object.__setattr__(blk, "period_selector", self.period_selector)   # shared, registered once
        self.residual_dropout = nn.Dropout(self.dropout)
        self.pre_embedding_dropout = nn.Dropout(self.dropout)
        # the time projection starts as "repeat the last observed step" (:1451-1455)
        self.forecast_time_proj = _zero_linear(self.input_len, self.pred_len)
        if self.pred_len > 0:
            with torch.no_grad():
                self.forecast_time_proj.weight[:, -1].fill_(1.0)

        # -- everything below is built on the first forward (reference :1514-1849); None until then
        for name in ("layer_norm", "embedding", "mu_head", "sigma_head", "series_embedding", "static_proj",
                     "static_norm", "context_norm", "context_proj", "context_coeff", "temporal_context",
                     "late_bias_norm", "late_bias_head", "pre_embedding_norm"):
            setattr(self, name, None)
        self.register_parameter("late_bias_gate", None)
        self.register_buffer("min_sigma_vector", None)
        if min_sigma_vector is not None:
            self.min_sigma_vector = torch.as_tensor(min_sigma_vector, dtype=torch.float32).reshape(1, 1, -1)
        self.embedding_time_features: Optional[int] = None
        self.output_dim: Optional[int] = None
        self.input_channels: Optional[int] = None
        self._static_in_features: Optional[int] = None
        self._static_out_dim = 0
        self._series_id_vocab: Optional[int] = None
        self._series_id_reference: Optional[torch.Tensor] = None
        self._last_head_backend = "torch"
        self._last_embed_backend = "torch"
        self._defer_checks = False
        self._pending_bad = None

    # ---- lazy construction ------------------------------------------------------
    def _bind_static(self, series_static: Optional[torch.Tensor], n_series: int, ref: torch.Tensor) -> int:
        """Static covariates -> ``static_proj`` (+ ``static_norm``); returns the projected width (0 = unused)."""
        if series_static is not None:
            if series_static.ndim not in (2, 3):
                raise ValueError("series_static must have shape [N, F] or [B, N, F]")
            rows, feat = series_static.shape[-2], int(series_static.shape[-1])
            if rows != n_series:
                raise ValueError("series_static must align with the number of input series")
            if feat <= 0:
                raise ValueError("series_static must have at least one feature")
            if self.static_proj is None:
                width = feat if self.static_proj_dim is None else self.static_proj_dim
                self.static_proj = nn.Linear(feat, width)
                self.static_norm = nn.LayerNorm(width) if self.static_layernorm else nn.Identity()
                self._static_in_features = feat
            elif self.static_proj.in_features != feat:
                raise ValueError("series_static feature dimension changed between calls")
        if self.static_proj is None:
            return 0
        self.static_proj = _place(self.static_proj, ref)
        if self.static_norm is not None:
            self.static_norm = _place(self.static_norm, ref)
        return int(self.static_proj.out_features)

    def _bind_ids(self, series_ids: Optional[torch.Tensor], n_series: int, ref: torch.Tensor) -> int:
        """Series identifiers -> ``series_embedding`` and the remembered id row; returns the embedding width."""
        if self.id_embed_dim <= 0:
            return 0
        row = None
        if series_ids is not None:
            if series_ids.ndim not in (1, 2):
                raise ValueError("series_ids must have shape [N] or [B, N]")
            row = (series_ids if series_ids.ndim == 1 else series_ids[0]).to(torch.long)
            if row.numel() != n_series:
                raise ValueError("series_ids length must match number of series")
        vocab = (lambda t: int(t.max().item()) + 1 if t.numel() else n_series)
        if self.series_embedding is None:
            if row is None:
                row = torch.arange(n_series, dtype=torch.long, device=ref.device)
            self.series_embedding = nn.Embedding(vocab(row), self.id_embed_dim)
        elif row is not None and not self._defer_checks and vocab(row) > int(self.series_embedding.num_embeddings):
            raise ValueError("series_ids vocabulary expanded between calls")     # (one host read; skipped in a capture)
        self.series_embedding = _place(self.series_embedding, ref)
        if row is not None:
            self._series_id_reference = row.to(ref.device)
        elif self._series_id_reference is None:
            self._series_id_reference = torch.arange(n_series, dtype=torch.long, device=ref.device)
        if self._series_id_reference.numel() != n_series:
            raise ValueError("series identifier count changed between calls")
        self._series_id_vocab = int(self.series_embedding.num_embeddings)
        return int(self.series_embedding.embedding_dim)

    def _ensure_embedding(self, x, x_mark=None, series_static=None, series_ids=None) -> None:
        """Build / resize / move every lazily constructed layer for this call's input sizes (:1514-1849)."""
        n_series, time_dim = x.shape[-1], (0 if x_mark is None else x_mark.shape[-1])
        for attr, val, msg in (("input_channels", n_series, "Number of series changed between calls"),
                               ("d_model", self.requested_d_model, "d_model changed between calls")):
            have = getattr(self, attr)
            if have is not None and have != val:
                raise ValueError(msg)
            setattr(self, attr, val)
        if self.embedding_time_features not in (None, time_dim):
            raise ValueError("Temporal feature dimension changed between calls")
        self.d_ff = self.requested_d_ff if self.requested_d_ff is not None else self.d_model
        self._static_out_dim = self._bind_static(series_static, n_series, x)
        dims = _Dims(n_series, time_dim, self._static_out_dim, self._bind_ids(series_ids, n_series, x),
                     self._out_steps, self.d_model)

        for spec in _LAZY_TABLE:
            mod = getattr(self, spec.name)
            if not spec.wanted(self, dims):
                mod = None
            else:
                if mod is None or not spec.fits(self, dims)(mod):
                    mod = spec.make(self, dims)
                mod = _place(mod, x)
            setattr(self, spec.name, mod)
        # the late-bias gate is a bare Parameter next to its head (:1836-1847)
        if self.late_bias_head is None:
            self.late_bias_gate = None
        else:
            gate, want = self.late_bias_gate, (1, dims.steps, 1)
            if isinstance(gate, nn.Parameter) and tuple(gate.shape) == want:
                gate.data = gate.data.to(x.device, torch.float32)
            else:
                self.late_bias_gate = nn.Parameter(torch.full(want, 0.05, device=x.device))
        self.pre_embedding_dropout.to(x.device)
        self.forecast_time_proj = _place(self.forecast_time_proj, x)

        floor = self.min_sigma_vector
        if isinstance(floor, torch.Tensor) and floor.numel() > 0:
            if int(floor.shape[-1]) < n_series:
                raise ValueError("min_sigma_vector length does not match number of series")
            self.min_sigma_vector = floor[..., :n_series]
        self.embedding_time_features = time_dim
        self.output_dim = n_series

    def _dispersion_floor_from_ref(self, ref: torch.Tensor) -> torch.Tensor:
        floor = self.min_sigma_vector
        if isinstance(floor, torch.Tensor) and floor.numel() > 0:
            return floor.to(device=ref.device, dtype=ref.dtype).expand_as(ref).clone()
        return torch.full_like(ref, self.min_sigma)

    # ---- series context: [Bc, N, ctx] rows the context / late-bias maps read ----------------
    def _context_rows(self, window, series_static, series_ids) -> Optional[torch.Tensor]:
        """Static projection and id embedding side by side, through ``context_norm`` (:1883-1956).  Every op is
        row-wise, so when the static features / ids are shared by the batch (2-D / 1-D inputs, the pipeline's
        case) the rows are built once (Bc = 1) and broadcast by their consumers."""
        B, _, N = window.shape
        shared = ((series_static is None or series_static.ndim == 2)
                  and (series_ids is None or series_ids.ndim == 1 or series_ids.size(0) == 1))
        Bc = 1 if shared else B
        cols = []
        if self.static_proj is not None and series_static is not None:
            if series_static.ndim == 3 and series_static.size(0) != B:
                raise ValueError("series_static batch dimension must match input batch size")
            st = series_static if series_static.ndim == 3 else series_static.unsqueeze(0).expand(Bc, -1, -1)
            proj = self.static_proj(st.to(device=window.device, dtype=window.dtype, non_blocking=window.is_cuda))
            cols.append(proj if self.static_norm is None else _norm(self.static_norm, proj))
        if self.series_embedding is not None and self.id_embed_dim > 0:
            if series_ids is not None:
                ids = series_ids.view(1, -1) if series_ids.ndim == 1 else series_ids
                if ids.size(0) not in (1, B):
                    raise ValueError("series_ids batch dimension does not match input")
                if ids.size(1) != N:
                    raise ValueError("series_ids length must match number of series")
                ids = ids.to(device=window.device, dtype=torch.long)
                self._series_id_reference = ids[0].detach().clone()
            else:
                remembered = self._series_id_reference
                if remembered is not None and remembered.numel() != N:
                    raise ValueError("Stored series identifiers do not match input dimension")
                ids = (torch.arange(N, device=window.device) if remembered is None else remembered.to(window.device)).view(1, N)
            cols.append(self.series_embedding(ids.expand(Bc, -1) if ids.size(0) != Bc else ids))
        if not cols:
            return None
        rows = cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)
        return rows if self.context_norm is None else _norm(self.context_norm, rows)

    # ---- embedding front end ---------------------------------------------------------------
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

    def _embed(self, window, mark, rows) -> torch.Tensor:
        """``embedding(window + temporal context + constant bias)`` (:1958-2020).  On the HIP path the two context
        terms are pushed through the value embedding's weight instead (``x W^T + (basis coeff^T + cb) W^T``): one pass
        over the window (``ftn_embed_forward``) and the [B, L, N] context tensor is never formed."""
        coeff = bias = None
        if rows is not None:
            if self.context_coeff is not None and self.temporal_context is not None and self.use_zero_mean_context:
                coeff = self.context_coeff(rows.to(self.context_coeff.weight.dtype))
            if self.context_proj is not None and self.use_constant_context_bias:
                bias = self.context_proj(rows.to(self.context_proj.weight.dtype)).squeeze(-1)
        if self._hip_embed_ok(window):
            from .. import runtime

            emb, L = self.embedding, window.size(1)
            w = emb.value_embedding.weight.detach()
            add = emb.aux_term(window[:1], mark) + emb.value_embedding.bias.detach()
            if coeff is not None:
                add = add + self.temporal_context.project(coeff.detach().float(), L, w)
            if bias is not None:
                add = add + (bias.detach().float() @ w.t()).unsqueeze(1)
            ln = None
            if emb.embed_norm_mode == "layer":
                ln = (emb.norm.weight.detach().float().contiguous(), emb.norm.bias.detach().float().contiguous(), emb.norm.eps)
            self._last_embed_backend = "hip"
            seq = runtime.embed_forward(window, w, add.detach().float().contiguous(), ln)
        else:
            feats = window
            if coeff is not None:
                signal = self.temporal_context(coeff, window.size(1))          # HIP LRTC kernel on ROCm tensors
                if signal.ndim != 3 or signal.shape[1:] != window.shape[1:] or signal.size(0) not in (1, window.size(0)):
                    raise RuntimeError("Temporal context must align with the [B, L, N] input")
                feats = feats + signal.to(feats.dtype)
            if bias is not None:
                feats = feats + bias.to(feats.dtype).unsqueeze(1)
            seq = self.embedding(feats, mark)
        if seq.ndim != 3 or seq.size(1) != self.input_len or seq.size(-1) != self.d_model:
            raise RuntimeError("Embedding output must have shape [B, input_len, d_model]")
        return seq

    # ---- block stack: x <- LayerNorm(x + dropout(block(x) - x)), one shared LayerNorm (:2022-2061) --------------
    def _stack(self, seq: torch.Tensor) -> torch.Tensor:
        self.period_selector = self.period_selector.to(device=seq.device, dtype=seq.dtype)
        for blk in self.blocks:
            object.__setattr__(blk, "period_selector", self.period_selector)
        recompute = self.use_checkpoint and torch.is_grad_enabled()
        plain_eval = not recompute and not (self.training and self.dropout > 0.0)
        for blk in self.blocks:
            if plain_eval:       # dropout is the identity: residual + LayerNorm ride in the block's last kernel
                seq = blk(seq, post_norm=self.layer_norm)
            else:
                new = checkpoint(blk, seq, use_reentrant=False) if recompute else blk(seq)
                seq = _norm(self.layer_norm, seq + self.residual_dropout(new - seq))
        return seq

    # ---- heads -----------------------------------------------------------------------------------
    def _hip_heads_ok(self, seq: torch.Tensor, window: torch.Tensor) -> bool:
        params = (self.forecast_time_proj.weight, self.mu_head.weight, self.sigma_head.weight)
        return (seq.is_cuda and seq.dtype == torch.float32 and window.dtype == torch.float32
                and not (torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in params)))
                and self.d_model % 4 == 0 and self.d_model <= 128
                and window.stride(2) == 1 and window.stride(1) == window.size(2))

    def _heads(self, seq, window, rows, steps: int):
        """Time projection L -> steps, then ``rate = softplus(mu + last observed values (+ late bias)) + 1e-6`` and
        ``dispersion = softplus(sigma) + floor + 1e-6`` (:2063-2102)."""
        B, L, N = window.shape
        hist = min(steps, L)
        late = None
        if (rows is not None and self.late_bias_head is not None and self.late_bias_norm is not None
                and isinstance(self.late_bias_gate, nn.Parameter)):
            head = self.late_bias_head
            lb = head(_norm(self.late_bias_norm, rows.to(device=head.weight.device, dtype=head.weight.dtype)))
            late = self.late_bias_gate.to(lb) * lb.transpose(1, 2)                          # [Bc, steps, N]
        wt, bt = self.forecast_time_proj.weight, self.forecast_time_proj.bias
        if steps != self.pred_len:
            wt, bt = wt[-steps:], bt[-steps:]
        if self._hip_heads_ok(seq, window):
            from .. import runtime

            # one batched GEMM straight into [B, steps, d_model] (no permute copies), then ftn_head_forward
            hidden = torch.baddbmm(bt.detach().view(1, -1, 1), wt.detach().unsqueeze(0).expand(B, -1, -1), seq.detach())
            floor = self.min_sigma_vector
            floor_vec = None
            if isinstance(floor, torch.Tensor) and floor.numel() > 0:
                floor_vec = floor.to(device=seq.device, dtype=torch.float32).reshape(-1).contiguous()
            rate, dispersion, bad = runtime.head_forward(
                hidden.contiguous(), self.mu_head.weight.detach(), self.mu_head.bias.detach(),
                self.sigma_head.weight.detach(), self.sigma_head.bias.detach(), window[:, -hist:, :], hist,
                None if late is None else late.detach().float().contiguous(), floor_vec, self.min_sigma)
            self._last_head_backend = "hip"
            self._pending_bad = bad
            if not self._defer_checks:
                self.check_outputs()
            return rate, dispersion
        hidden = torch.matmul(wt, seq) + bt.view(1, -1, 1)                                  # [B, steps, d_model]
        last = window[:, -hist:, :]
        if hist < steps:                                       # recursive mode / short windows: repeat the last step
            last = torch.cat([last, last[:, -1:, :].expand(-1, steps - hist, -1)], dim=1)
        pre = self.mu_head(hidden) + last.to(window.dtype)
        if late is not None:
            pre = pre + late.to(pre.dtype)
        soft = lambda t: F.softplus(t.float(), beta=1.0, threshold=20).to(t.dtype)
        rate = soft(pre) + 1e-6
        spread = soft(self.sigma_head(hidden))
        dispersion = spread + self._dispersion_floor_from_ref(rate).to(spread.dtype) + 1e-6
        for name, val in (("rate", rate), ("dispersion", dispersion)):
            if not bool((torch.isfinite(val) & (val > 0)).all()):
                raise RuntimeError(f"Predicted {name} must be finite and strictly positive")
        if rate.shape != (B, steps, N) or dispersion.shape != (B, steps, N):
            raise RuntimeError("Predicted rate/dispersion have incorrect shape")
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
        for bit, name in ((1, "rate"), (2, "dispersion")):
            if flag & bit:
                raise RuntimeError(f"Predicted {name} must be finite and strictly positive")

    # ---- forward ------------------------------------------------------------------
    def forward(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None,
                series_static: Optional[torch.Tensor] = None,
                series_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        out = self._forward_once(x, x_mark, series_static, series_ids)
        if getattr(self, "_range_retry", False):                # check_outputs(): a block left the f16x2 range
            self._range_retry = False
            out = self._forward_once(x, x_mark, series_static, series_ids)
        return out

    def _forward_once(self, x, x_mark, series_static, series_ids) -> Tuple[torch.Tensor, torch.Tensor]:
        L = self.input_len
        if x.dim() != 3:
            raise ValueError("TimesNet expects input shaped [B, T, N]")
        if x.size(1) < L:
            raise ValueError(f"Input sequence length {x.size(1)} is shorter than required input_len {L}")
        if x_mark is not None and tuple(x_mark.shape[:2]) != tuple(x.shape[:2]):
            raise ValueError("x_mark must share batch/time dimensions with x")
        window = x.narrow(1, x.size(1) - L, L)                    # the last input_len steps (a view)
        mark = None if x_mark is None else x_mark.narrow(1, x_mark.size(1) - L, L)
        self._ensure_embedding(window, mark, series_static, series_ids)
        rows = self._context_rows(window, series_static, series_ids)
        seq = self._stack(self._embed(window, mark, rows))
        return self._heads(seq, window, rows, self._out_steps)
