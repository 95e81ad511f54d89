"""CPU oracle for the TimesBlock forward path — TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional* restatement (stock torch ops on CPU
tensors, fp32 or fp64) of the algorithm in the reference's
``src/timesnet_forecast/models/timesnet.py``.  It exists so that the HIP path
can be checked on a box where the reference cannot travel.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``flow-timesnet_amd/``) never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
in the build container, runs it on seeded inputs and commits inputs+outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors and against the reference's own known-answer tests
(``tests/test_fft_period_selector.py:14-70,105-118``,
``tests/test_times_block.py:101-136``,
``tests/test_timesblock_vectorized.py:111-129``).

Each function cites the reference lines it follows (paths relative to
``/root/reference/src/timesnet_forecast/models/timesnet.py``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# Period selector                                     timesnet.py:52-159
# --------------------------------------------------------------------------
@dataclass
class SelectResult:
    freq_idx: List[int]            # kept frequency bins (score order)
    periods: List[int]             # kept periods (score order, dups allowed)
    amps: torch.Tensor             # [B, K'] channel-median amplitude at kept bins
    amp_mean: torch.Tensor         # [F] batch mean of channel-median (pre DC kill)
    median: torch.Tensor           # [B, F] channel-median amplitude
    topk_gap: float = float("inf")  # relative gap between k-th and (k+1)-th score


def channel_median_spectrum(x: torch.Tensor) -> torch.Tensor:
    """|rfft_t x|, then the *lower* median over channels.  :108-111"""
    spec = torch.fft.rfft(x, dim=1)                    # [B, F, C] complex
    amp = spec.abs()
    srt, _ = torch.sort(amp, dim=2)
    C = x.shape[2]
    return srt[:, :, (C - 1) // 2]                      # lower median == torch.median


def period_select(
    x: torch.Tensor, k_periods: int, pmax: int, min_period_threshold: int = 1
) -> SelectResult:
    """FFTPeriodSelector.forward restated.  :64-159 (ctor clamps :59-62)."""
    k_cfg = max(0, int(k_periods))
    pmax = max(1, int(pmax))
    min_thr = min(pmax, max(1, int(min_period_threshold)))
    B, L, C = x.shape
    empty = SelectResult([], [], x.new_zeros(B, 0), x.new_zeros(0), x.new_zeros(B, 0))
    if k_cfg <= 0 or L <= 1 or C <= 0 or B <= 0:                      # :89-90
        return empty
    xf = x.float() if x.dtype in (torch.float16, torch.bfloat16) else x  # :92-94
    med = channel_median_spectrum(xf)                                  # :109-111
    amp_mean = med.mean(dim=0)                                         # :112
    Fbins = amp_mean.numel()
    if Fbins <= 1:
        return empty
    scores = amp_mean.clone()
    scores[0] = float("-inf")                                          # :120
    k = min(k_cfg, Fbins - 1)                                          # :122-123
    if k <= 0:
        return empty
    idx = torch.arange(Fbins, dtype=torch.float32)
    scores = scores - (1e-8 * torch.log1p(idx)).to(scores.dtype)       # :128-130
    order = _topk_lowest_index_first(scores, k)                        # :131
    srt = torch.sort(scores, descending=True).values
    gap = float("inf")
    if Fbins - 1 > k:
        denom = max(abs(float(srt[k - 1])), 1e-30)
        gap = float(srt[k - 1] - srt[k]) / denom
    safe = [max(int(i), 1) for i in order]                             # :132
    hi = min(pmax, max(1, L - 1))                                      # :138
    lo = min_thr                                                       # :139
    if hi < lo:
        return empty
    periods = [min(max((L + i - 1) // i, lo), hi) for i in safe]       # :144-145
    keep = [j for j, p in enumerate(periods) if (L + p - 1) // p >= 2]  # :147-148
    if not keep:
        return empty
    kept_idx = [safe[j] for j in keep]
    amps = med[:, kept_idx].to(x.dtype)                                # :133-135,159
    return SelectResult(kept_idx, [periods[j] for j in keep], amps, amp_mean, med, gap)


def _topk_lowest_index_first(scores: torch.Tensor, k: int) -> List[int]:
    """Descending top-k; ties resolved lowest-index-first (torch.topk's tie
    order is implementation-defined, SURVEY §7 'Exact top-k')."""
    vals = scores.tolist()
    order = sorted(range(len(vals)), key=lambda i: (-vals[i], i))
    return order[:k]


# --------------------------------------------------------------------------
# Period grouper (env flags unset)                    timesnet.py:513-557
# --------------------------------------------------------------------------
@dataclass
class GroupResult:
    periods: List[int] = field(default_factory=list)   # [G] ascending
    pad: List[int] = field(default_factory=list)
    cycles: List[int] = field(default_factory=list)
    mapping: List[int] = field(default_factory=list)   # [K] -> group or -1


def period_group(
    periods: Sequence[int], L: int, min_period: Optional[int], max_period: Optional[int]
) -> GroupResult:
    """Duplicate-merge grouping.  :517-551 (filter), :453-477 (order/mapping)."""
    K = len(periods)
    res = GroupResult(mapping=[-1] * K)
    cand = []
    for j, p in enumerate(periods):
        p = int(p)
        if p <= 0:
            continue
        if min_period is not None and p < min_period:
            continue
        if max_period is not None and p > max_period:
            continue
        pad = (-L) % p                                                  # :531
        cyc = (L + pad) // p                                            # :532-533
        if cyc < 2:                                                     # :534
            continue
        cand.append((j, p, pad, cyc))
    if not cand:
        return res
    uniq = sorted({p for _, p, _, _ in cand})                           # :551,:453-458
    gid = {p: g for g, p in enumerate(uniq)}
    for j, p, pad, cyc in cand:
        res.mapping[j] = gid[p]
    for p in uniq:
        pad = (-L) % p
        res.periods.append(p)
        res.pad.append(pad)
        res.cycles.append((L + pad) // p)
    return res


def group_weights(amps: torch.Tensor, mapping: Sequence[int], G: int) -> torch.Tensor:
    """softmax over valid candidates in fp32, scatter-add into groups. :992-1009"""
    valid = [j for j, g in enumerate(mapping) if g >= 0]
    sm = F.softmax(amps[:, valid].float(), dim=1).to(amps.dtype)       # :1000
    w = amps.new_zeros(amps.shape[0], G)
    for col, j in enumerate(valid):
        w[:, mapping[j]] += sm[:, col]
    return w


# --------------------------------------------------------------------------
# Inception                                           timesnet.py:560-654
# --------------------------------------------------------------------------
def _act(z: torch.Tensor, name: str) -> torch.Tensor:
    return F.relu(z) if name == "relu" else F.gelu(z)   # nn.GELU() == erf form :639-643


def bottleneck_mid(in_ch: int, out_ch: int, ratio: float) -> Optional[int]:
    """None when the branch is a single conv (ratio ~ 1).  :575,:582-585"""
    if math.isclose(ratio, 1.0, rel_tol=1e-9, abs_tol=1e-9):
        return None
    return max(1, int(math.ceil(min(in_ch, out_ch) / float(ratio))))


def inception_block(
    u: torch.Tensor, P: Params, prefix: str, kernel_set: Sequence[Tuple[int, int]],
    act: str,
) -> torch.Tensor:
    """InceptionBlock.forward on an NCHW grid; dropout is identity (eval). :645-654"""
    feats = []
    for j, (kh, kw) in enumerate(kernel_set):
        h = u
        i = 0
        while f"{prefix}paths.{j}.branch.{i}.weight" in P:              # :578-590
            w = P[f"{prefix}paths.{j}.branch.{i}.weight"]
            b = P[f"{prefix}paths.{j}.branch.{i}.bias"]
            pad = (w.shape[2] // 2, w.shape[3] // 2)                    # :574
            h = F.conv2d(h, w, b, padding=pad)
            i += 1
        feats.append(h)
    z = torch.cat(feats, dim=1)                                         # :650
    z = F.conv2d(z, P[f"{prefix}proj.weight"], P[f"{prefix}proj.bias"])  # :651
    z = _act(z, act)                                                    # :652
    if f"{prefix}res_proj.weight" in P:                                 # :634-637
        res = F.conv2d(u, P[f"{prefix}res_proj.weight"], P[f"{prefix}res_proj.bias"])
    else:
        res = u
    return z + res                                                      # :654


def inception_stack(grid: torch.Tensor, P: Params, kernel_set, act: str) -> torch.Tensor:
    """Sequential(InceptionBlock, act, InceptionBlock).  :744-762"""
    h = inception_block(grid, P, "0.", kernel_set, act)
    h = _act(h, act)
    return inception_block(h, P, "2.", kernel_set, act)


# --------------------------------------------------------------------------
# TimesBlock                                          timesnet.py:767-818, 955-1101
# --------------------------------------------------------------------------
@dataclass
class BlockAux:
    sel: Optional[SelectResult] = None
    groups: Optional[GroupResult] = None
    weights: Optional[torch.Tensor] = None
    deltas: List[torch.Tensor] = field(default_factory=list)


def period_delta(x: torch.Tensor, period: int, P: Params, kernel_set, act: str) -> torch.Tensor:
    """One group's residual: fold, inception, minus grid, unfold, crop. :1041-1069"""
    B, L, C = x.shape
    pad = (-L) % period
    xp = F.pad(x.permute(0, 2, 1), (0, pad))                            # tail pixels are live zeros
    grid = xp.reshape(B, C, (L + pad) // period, period)
    out = inception_stack(grid, P, kernel_set, act)
    return (out - grid).reshape(B, C, L + pad)[..., :L].permute(0, 2, 1)


def timesblock_forward(
    x: torch.Tensor, P: Params, kernel_set: Sequence[Tuple[int, int]], act: str,
    k_periods: int, pmax: int, min_period_threshold: int = 1,
    periods: Optional[Sequence[int]] = None, amps: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, BlockAux]:
    """TimesBlock.forward, default (bucketed) path.  ``periods``/``amps`` inject
    a stub selector the way the reference tests do (tests/test_times_block.py:14-30)."""
    aux = BlockAux()
    B, L, C = x.shape
    if periods is None:
        aux.sel = period_select(x, k_periods, pmax, min_period_threshold)
        periods, amps = aux.sel.periods, aux.sel.amps
    if len(periods) == 0:                                               # :796-797
        return x, aux
    if amps.dim() == 1:
        amps = amps.view(1, -1).expand(B, -1)                           # :993-994
    elif amps.shape[0] == 1 and B > 1:
        amps = amps.expand(B, -1)
    pmax_c = max(1, int(pmax))
    min_c = min(pmax_c, max(1, int(min_period_threshold)))
    aux.groups = period_group(periods, L, min_c, pmax_c)                # :975-984
    G = len(aux.groups.periods)
    if G == 0:                                                          # :989-990
        return x, aux
    aux.weights = group_weights(amps, aux.groups.mapping, G)
    combined = torch.zeros_like(x)
    for g, p in enumerate(aux.groups.periods):
        d = period_delta(x, p, P, kernel_set, act)
        aux.deltas.append(d)
        combined = combined + d * aux.weights[:, g].view(B, 1, 1)       # :1075-1092
    return x + combined, aux                                            # :818


# --------------------------------------------------------------------------
# LowRankTemporalContext                              timesnet.py:1340-1371
# --------------------------------------------------------------------------
def lrtc_basis(L: int, R: int, dtype=torch.float32) -> torch.Tensor:
    steps = torch.arange(L, dtype=dtype).unsqueeze(1)
    freqs = torch.arange(1, R + 1, dtype=dtype).unsqueeze(0)
    basis = torch.cos(math.pi / float(L) * (steps + 0.5) * freqs)       # :1346
    basis = basis - basis.mean(dim=0, keepdim=True)                     # :1347
    norm = torch.linalg.norm(basis, dim=0, keepdim=True)                # :1348
    return basis / norm.clamp_min(torch.finfo(dtype).eps)               # :1349-1350


def lrtc_forward(coeff: torch.Tensor, L: int, scale: float) -> torch.Tensor:
    """coeff[B,N,R] -> ctx[B,L,N].  :1362-1371"""
    basis = lrtc_basis(L, coeff.shape[-1], coeff.dtype)
    ctx = torch.einsum("lr,bnr->bln", basis, coeff)                     # :1368
    ctx = ctx - ctx.mean(dim=1, keepdim=True)                           # :1369
    return ctx * torch.as_tensor(scale, dtype=coeff.dtype)             # :1370-1371


# ---------------------------------------------------------------------------
# Model shell pieces around the block stack
# ---------------------------------------------------------------------------
def block_post_norm(x: torch.Tensor, new: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                    eps: float = 1e-5) -> torch.Tensor:
    """Per-block epilogue of TimesNet.forward in eval mode (reference models/timesnet.py:2050-2058):
    ``LayerNorm(seq + dropout(updated - seq))`` with dropout = identity."""
    return F.layer_norm(x + (new - x), (x.shape[-1],), gamma, beta, eps)


def model_heads(hidden: torch.Tensor, w_mu: torch.Tensor, b_mu: torch.Tensor, w_sigma: torch.Tensor,
                b_sigma: torch.Tensor, tail: torch.Tensor, late: Optional[torch.Tensor] = None,
                floor=1e-3) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rate / dispersion heads of TimesNet.forward (reference :2066-2102).  ``hidden`` [B,S,D] is the
    time-projected feature map, ``tail`` [B,S,N] the (edge-padded) last observed values (:2008-2014),
    ``late`` the gated late bias [1|B,S,N] (:2080-2090), ``floor`` min_sigma or min_sigma_vector."""
    pre = F.linear(hidden, w_mu, b_mu) + tail
    if late is not None:
        pre = pre + late
    rate = F.softplus(pre.float(), beta=1.0, threshold=20) + 1e-6
    sig = F.softplus(F.linear(hidden, w_sigma, b_sigma).float(), beta=1.0, threshold=20)
    floor_t = floor if isinstance(floor, torch.Tensor) else torch.full_like(sig, float(floor))
    return rate, sig + floor_t + 1e-6


def embed_front(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, aux_term: torch.Tensor,
                signal: Optional[torch.Tensor] = None, cbias: Optional[torch.Tensor] = None,
                ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> torch.Tensor:
    """Context front-end + DataEmbedding.forward (reference :1958-1996, :1283-1325):
    ``feats = x + signal + cbias[:, None, :]`` (temporal context, constant context bias),
    ``value = Linear(feats)``, ``out = value + aux_term`` where ``aux_term`` is ``gate * LayerNorm(aux)``
    ("decoupled") or ``aux`` itself; ``ln`` = the "layer" mode's LayerNorm over the sum."""
    feats = x
    if signal is not None:
        feats = feats + signal
    if cbias is not None:
        feats = feats + cbias.unsqueeze(1)
    out = F.linear(feats, w, b) + aux_term
    if ln is not None:
        out = F.layer_norm(out, (out.shape[-1],), ln[0], ln[1], ln[2])
    return out
