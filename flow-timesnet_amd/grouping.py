"""Host mirror of the reference's ``PeriodGrouper`` / ``PeriodGroupResult``
(reference ``models/timesnet.py:162-557``), including the env-flag variants
``TIMES_PERIOD_MAX_UNIQ`` and ``TIMES_PERIOD_BINNING`` (depth schedules such as
``"0:4,2:2,default:3"``).

The default (flags unset) grouping also exists on the device
(``ftn_period_finalize``); this host version serves stub selectors, the flag
variants and CPU tensors.  K <= 16 candidates, so it works on Python scalars and
only builds tensors for the result.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch


# ---- env schedule parsing ----------------------------------------- :162-272
def _resolve_scheduled_value(raw: Optional[str], depth: Optional[int]) -> Optional[str]:
    """``"4"`` | ``"0:4,2:2,default:3"`` | ``"0=4,*=3"`` -> the entry for ``depth``:
    exact key, else nearest lower key, else default, else last bare token, else
    the smallest key's value."""
    if raw is None or not raw.strip():
        return None
    toks = [t.strip() for t in raw.strip().split(",") if t.strip()]
    if not toks:
        return None
    bare: List[str] = []
    dflt: List[str] = []
    keyed: Dict[int, str] = {}
    for tok in toks:
        sep = ":" if ":" in tok else ("=" if "=" in tok else None)
        if sep is None:
            bare.append(tok)
            continue
        k, v = (s.strip() for s in tok.split(sep, 1))
        if not v:
            continue
        k = k.lower()
        if k in ("default", "*"):
            dflt.append(v)
        else:
            try:
                keyed[int(k)] = v
            except ValueError:
                pass
    if depth is not None and keyed:
        if depth in keyed:
            return keyed[depth]
        lower = [k for k in keyed if k <= depth]
        if lower:
            return keyed[max(lower)]
    if dflt:
        return dflt[-1]
    if bare:
        return bare[-1]
    if keyed:
        return keyed[min(keyed)]
    return toks[-1]


def _resolve_scheduled_int(raw: Optional[str], depth: Optional[int]) -> Optional[int]:
    v = _resolve_scheduled_value(raw, depth)
    if v is None:
        return None
    try:
        n = int(float(v))
    except ValueError:
        return None
    return n if n > 0 else None


def _resolve_log_binning_base(raw: Optional[str], depth: Optional[int]) -> Optional[float]:
    v = _resolve_scheduled_value(raw, depth)
    if v is None:
        return None
    text = v.strip().lower()
    if not text or text in ("off", "false", "0", "none"):
        return None
    words = ("log", "logscale", "logarithmic")
    base: Optional[float] = None
    if ":" in text:
        head, tail = (s.strip() for s in text.split(":", 1))
        try:
            base = float(tail if head in words else head)
        except ValueError:
            base = None
    elif text in words:
        base = 2.0
    else:
        try:
            base = float(text)
        except ValueError:
            base = None
    if base is None:
        base = 2.0
    return float(base) if base > 1.0 else None


@dataclass
class PeriodGroupResult:
    periods: torch.Tensor
    pad_lengths: torch.Tensor
    cycles: torch.Tensor
    logits: torch.Tensor
    mapping: torch.Tensor
    valid_mask: torch.Tensor
    canonical_indices: torch.Tensor


class PeriodGrouper:
    """Same constructor and ``group()`` contract as the reference (:289-299, :513)."""

    def __init__(self, periods: torch.Tensor, amplitudes: torch.Tensor, seq_len: int, *,
                 min_period: Optional[int] = None, max_period: Optional[int] = None,
                 block_index: Optional[int] = None, freq_indices: Optional[torch.Tensor] = None) -> None:
        self.periods = periods.view(-1)
        amp = amplitudes
        if amp.dim() == 1:
            amp = amp.unsqueeze(0)
        if amp.dim() != 2:
            raise ValueError("amplitudes must have shape [B, K] or [K]")
        if amp.size(1) != self.periods.numel():
            raise ValueError("amplitudes second dimension must match number of period candidates")
        self.amplitudes = amp
        self.seq_len = int(seq_len)
        self.device = self.periods.device
        self.batch = amp.size(0)
        self.amp_dtype = amp.dtype
        self.period_dtype = self.periods.dtype
        self.min_period = int(min_period) if min_period is not None else None
        self.max_period = int(max_period) if max_period is not None else None
        self.block_index = int(block_index) if block_index is not None else None
        self.freq_indices = freq_indices
        self.max_unique = _resolve_scheduled_int(os.getenv("TIMES_PERIOD_MAX_UNIQ"), self.block_index)
        self.log_base = _resolve_log_binning_base(os.getenv("TIMES_PERIOD_BINNING"), self.block_index)

    # -- helpers ---------------------------------------------------------
    def _empty(self) -> PeriodGroupResult:
        K = self.periods.numel()
        z = torch.zeros(0, dtype=self.period_dtype, device=self.device)
        return PeriodGroupResult(
            periods=z, pad_lengths=z, cycles=z,
            logits=torch.zeros(self.batch, 0, dtype=self.amp_dtype, device=self.amplitudes.device),
            mapping=torch.full((K,), -1, dtype=torch.long, device=self.device),
            valid_mask=torch.zeros(K, dtype=torch.bool, device=self.device),
            canonical_indices=torch.zeros(0, dtype=torch.long, device=self.device),
        )

    def _bucket(self, p: int) -> int:
        # floor(log_b p + 1e-6) evaluated in fp32 like the reference (:350-354)
        lv = torch.log(torch.tensor(float(p), dtype=torch.float32)) / math.log(self.log_base)
        return int(torch.floor(lv + 1e-6).item())

    def group(self) -> PeriodGroupResult:
        K = self.periods.numel()
        if K == 0:
            return self._empty()
        L = self.seq_len
        plist = [int(v) for v in self.periods.tolist()]
        cand = []                                   # (orig index, period, pad, cycles)
        for j, p in enumerate(plist):
            if p <= 0:
                continue
            if self.min_period is not None and p < self.min_period:
                continue
            if self.max_period is not None and p > self.max_period:
                continue
            pad = (-L) % p
            cyc = (L + pad) // p
            if cyc >= 2:
                cand.append((j, p, pad, cyc))
        if not cand:
            return self._empty()
        amp_sel = self.amplitudes[:, [c[0] for c in cand]]           # [B, M]
        keys = [c[1] if self.log_base is None else self._bucket(c[1]) for c in cand]
        uniq = sorted(set(keys))
        assign = [uniq.index(k) for k in keys]                        # :551

        def metadata(assign_now):
            info = []
            for gid in sorted(set(assign_now)):
                members = [m for m, a in enumerate(assign_now) if a == gid]
                ml = amp_sel[:, members]
                agg = torch.logsumexp(ml, dim=1)                      # :373
                best = 0 if len(members) == 1 else int(torch.argmax(ml.mean(dim=0)).item())
                cm = members[best]                                    # :374-378
                info.append(dict(id=gid, members=members, canon=cm, period=cand[cm][1], pad=cand[cm][2],
                                 cycles=cand[cm][3], logits=agg, score=float(agg.mean().item()),
                                 canon_index=cand[cm][0]))
            return info

        if self.max_unique is not None and len(set(assign)) > self.max_unique:   # :394-437
            info = metadata(assign)
            scores = torch.tensor([it["score"] for it in info], dtype=torch.float32)
            keep = torch.topk(scores, k=self.max_unique, largest=True).indices.tolist()
            keep_periods = torch.tensor([float(info[k]["period"]) for k in keep], dtype=torch.float32)
            new_assign = list(assign)
            for idx, it in enumerate(info):
                if idx in keep:
                    continue
                dist = torch.abs(keep_periods - float(it["period"]))
                target = info[keep[int(torch.argmin(dist).item())]]["id"]
                for m in it["members"]:
                    new_assign[m] = target
            assign = new_assign

        info = metadata(assign)
        info.sort(key=lambda it: (it["period"], it["canon_index"]))    # :453-458
        mapping = torch.full((K,), -1, dtype=torch.long, device=self.device)
        valid = torch.zeros(K, dtype=torch.bool, device=self.device)
        for c in cand:
            valid[c[0]] = True
        for new_idx, it in enumerate(info):
            for m in it["members"]:
                mapping[cand[m][0]] = new_idx
        mk = lambda key, dt: torch.tensor([it[key] for it in info], dtype=dt, device=self.device)
        return PeriodGroupResult(
            periods=mk("period", self.period_dtype), pad_lengths=mk("pad", self.period_dtype),
            cycles=mk("cycles", self.period_dtype),
            logits=torch.stack([it["logits"] for it in info], dim=1),
            mapping=mapping, valid_mask=valid, canonical_indices=mk("canon_index", torch.long),
        )
