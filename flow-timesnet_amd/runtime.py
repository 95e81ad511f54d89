"""Per-device HIP runtime state for the TimesBlock path: DFT twiddle tables,
LRTC bases, and the thin call wrappers that pass ``tensor.data_ptr()`` / the
current HIP stream into the C ABI.  Workspaces are allocated per call from
torch's caching allocator (stream- and graph-pool-safe; free after warm-up).

PyTorch is plumbing here (device memory + streams); every computation happens
in ``libflowtimes_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Dict, Tuple

import torch

from . import lib as _lib
from .lib import DESC_INTS, FTN_KMAX, FtnDesc, FtnPlan, check


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def _ptr_or_null(t):
    return None if t is None else t.data_ptr()


class DeviceState:
    """Caches that live as long as the process, one instance per CUDA device."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.tables: Dict[int, torch.Tensor] = {}
        self.bases: Dict[Tuple[int, int], torch.Tensor] = {}

    def dft_table(self, L: int) -> torch.Tensor:
        t = self.tables.get(L)
        if t is None:
            lib = _lib.load()
            nbytes = lib.ftn_dft_table_bytes(L)
            t = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            check(lib.ftn_dft_table_init(_ptr(t), L, _stream(self.device)), "ftn_dft_table_init")
            self.tables[L] = t
        return t

    def lrtc_basis(self, L: int, R: int) -> torch.Tensor:
        key = (L, R)
        t = self.bases.get(key)
        if t is None:
            lib = _lib.load()
            t = torch.empty(lib.ftn_lrtc_basis_floats(L, R), dtype=torch.float32, device=self.device)
            check(lib.ftn_lrtc_basis(_ptr(t), L, R, _stream(self.device)), "ftn_lrtc_basis")
            self.bases[key] = t
        return t


_states: Dict[int, DeviceState] = {}


def state(device: torch.device) -> DeviceState:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _states.get(idx)
    if st is None:
        st = DeviceState(torch.device("cuda", idx))
        _states[idx] = st
    return st


# ------------------------------------------------------------------ selector
class Selection:
    """Device-resident result of the period selector for one block call."""

    def __init__(self, desc: torch.Tensor, amps: torch.Tensor, weights: torch.Tensor, max_groups: int,
                 px_bound: int = 0) -> None:
        self.desc = desc          # int32 [DESC_INTS]
        self.amps = amps          # [B, FTN_KMAX]
        self.weights = weights    # [B, FTN_KMAX]
        self.max_groups = max_groups     # >= desc.n_groups
        self.px_bound = px_bound         # >= desc.total_px (0: generic worst case)
        self._host = None

    def host(self) -> FtnDesc:
        """Copy the descriptor to the host (synchronises the stream)."""
        if self._host is None:
            raw = self.desc.cpu().numpy().tobytes()
            self._host = FtnDesc.from_buffer_copy(raw)
        return self._host


def spectrum(x: torch.Tensor, xch=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """S1+S2 on the device: channel-median amplitude [B,F] and its fp64 batch sum [F].  ``xch`` (a ``ctypes.byref`` of
    an ``FtnExchange``): the sums are also stored into every rank's exchange buffer (``dist.IpcExchange``)."""
    lib = _lib.load()
    B, L, Cc = x.shape
    st = state(x.device)
    Fb = L // 2 + 1
    med = torch.empty(B, Fb, dtype=torch.float32, device=x.device)
    psum = torch.empty(Fb, dtype=torch.float64, device=x.device)
    nscr = int(lib.ftn_period_spectrum_scratch_bytes(B, L, Cc))      # > 0: the channel-tiled form of d_model > 64
    scratch = torch.empty(nscr, dtype=torch.uint8, device=x.device) if nscr else None
    check(lib.ftn_period_spectrum(_ptr(x), B, L, Cc, _ptr(st.dft_table(L)), _ptr(med), _ptr(psum),
                                  _stream(x.device), xch, _ptr_or_null(scratch)), "ftn_period_spectrum")
    return med, psum


ACT_DTYPE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _selector_bounds(lib, L: int, k: int, pmax: int, min_thr: int) -> Tuple[int, int]:
    mg = C.c_int(0)
    pxb = lib.ftn_selector_px_bound(L, int(k), int(pmax), int(min_thr), C.byref(mg))
    if pxb < 0:
        check(pxb, "ftn_selector_px_bound")
    return max(1, int(mg.value)), int(pxb)


def new_range_flag(device: torch.device) -> torch.Tensor:
    """One int32 the f16x2 kernels set when a value leaves the fp16 range (``include/flowtimes.h``, ABI 9).  By default
    it lives in pinned host memory, which the device writes directly (zero-copy): the host can then read it at any time
    without a copy or a synchronisation once the call's completion event has fired.  ``FTN_RANGE_FLAG=device`` keeps it
    in device memory (reading it then synchronises)."""
    if os.getenv("FTN_RANGE_FLAG", "host") == "device":
        return torch.zeros(1, dtype=torch.int32, device=device)
    return torch.zeros(1, dtype=torch.int32).pin_memory()


def fuse_stage_a(plan) -> bool:
    """Stage A can ride with the selector's finalize launch (bottleneck blocks; FTN_FUSE_STAGE_A=0 disables)."""
    return plan.mode == 0 and os.getenv("FTN_FUSE_STAGE_A", "1") != "0"


def stage_a_only(x: torch.Tensor, plan: FtnPlan, wblob: torch.Tensor, k: int, pmax: int, min_thr: int,
                 range_flag: Optional[torch.Tensor] = None):
    """Stage A of the block into a fresh workspace, without the selection (``ftn_period_finalize_stage_a`` with
    ``psum = NULL``): a batch-sharded run launches it between issuing the exchange of the partial sums and waiting
    for it.  Returns the token ``finalize(..., stage_a=..., pre=token)`` completes."""
    lib = _lib.load()
    B, L, _ = x.shape
    mg, pxb = _selector_bounds(lib, L, k, pmax, min_thr)
    need = lib.ftn_timesblock_workspace_bytes(C.byref(plan), B, L, mg, pxb)
    if need == 0:
        raise ValueError(f"ftn_timesblock_workspace_bytes rejected the shape (B={B}, L={L})")
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    check(lib.ftn_period_finalize_stage_a(None, 0, 0, None, B, L, int(k), int(pmax), int(min_thr), 0, 0, 0.0,
                                          None, None, None, _ptr(x), C.byref(plan), _ptr(wblob), mg, pxb, _ptr(ws),
                                          ws.numel(), _stream(x.device), _ptr_or_null(range_flag), None),
          "ftn_period_finalize_stage_a")
    return ws


def finalize(psum: torch.Tensor, b_total: int, med: torch.Tensor, L: int, k: int, pmax: int,
             min_thr: int, act_dtype: int = 0, max_unique: int = 0, log_base: float = 0.0,
             stage_a=None, pre=None, xch=None) -> Selection:
    """S3-S5 on the device.  ``psum`` is [F] or [nparts, F] (multi-GPU partial sums); ``act_dtype`` 1 / 2
    applies the reference's bf16 / fp16 roundings of scores, amplitudes and weights.

    ``stage_a=(x, plan, wblob)``: the block that consumes this selection is known, so its stage A
    (a = W_in1 x + b, independent of the selection) rides in the same launch
    (``ftn_period_finalize_stage_a``); the returned Selection then owns the block's workspace and
    ``timesblock_forward`` skips stage A.  ``pre`` = the workspace ``stage_a_only`` already filled: only S3-S5 and
    the descriptor copy remain."""
    lib = _lib.load()
    B = med.shape[0]
    dev = med.device
    nparts = 1 if psum.dim() == 1 else psum.shape[0]     # (with ``xch`` the parts are the exchange buffer's slots)
    desc = torch.empty(DESC_INTS, dtype=torch.int32, device=dev)
    amps = torch.empty(B, FTN_KMAX, dtype=torch.float32, device=dev)
    wts = torch.empty(B, FTN_KMAX, dtype=torch.float32, device=dev)
    mg, pxb = _selector_bounds(lib, L, k, pmax, min_thr)
    sel = Selection(desc, amps, wts, mg, pxb)
    if stage_a is not None and (pre is not None or fuse_stage_a(stage_a[1])):
        x, plan, wblob = stage_a[:3]
        range_flag = stage_a[3] if len(stage_a) > 3 else None
        if pre is None:
            need = lib.ftn_timesblock_workspace_bytes(C.byref(plan), B, L, sel.max_groups, sel.px_bound)
            if need == 0:
                raise ValueError(f"ftn_timesblock_workspace_bytes rejected the shape (B={B}, L={L})")
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        else:
            ws = pre
        check(lib.ftn_period_finalize_stage_a(_ptr(psum), nparts, int(b_total), _ptr(med), B, L, int(k), int(pmax),
                                              int(min_thr), int(act_dtype), int(max_unique or 0),
                                              float(log_base or 0.0), _ptr(desc), _ptr(amps), _ptr(wts),
                                              _ptr(x) if pre is None else None,
                                              C.byref(plan), _ptr(wblob), sel.max_groups, sel.px_bound, _ptr(ws),
                                              ws.numel(), _stream(dev), _ptr_or_null(range_flag), xch),
              "ftn_period_finalize_stage_a")
        sel.stage_a = (ws, x.data_ptr(), C.addressof(plan))
        return sel
    check(lib.ftn_period_finalize(_ptr(psum), nparts, int(b_total), _ptr(med), B, L, int(k), int(pmax),
                                  int(min_thr), int(act_dtype), int(max_unique or 0), float(log_base or 0.0), _ptr(desc),
                                  _ptr(amps), _ptr(wts), _stream(dev), xch),
          "ftn_period_finalize")
    return sel


def selection_from_host(desc_host: FtnDesc, weights: torch.Tensor, device: torch.device) -> Selection:
    """Upload a host-built descriptor + [B,G] group weights (stub selectors / env flags)."""
    import numpy as np

    raw = np.frombuffer(bytes(desc_host), dtype=np.int32).copy()
    desc = torch.from_numpy(raw).to(device)
    B, G = weights.shape
    w = torch.zeros(B, FTN_KMAX, dtype=torch.float32, device=device)
    w[:, :G] = weights.to(device=device, dtype=torch.float32)
    sel = Selection(desc, w, w, max(1, int(desc_host.n_groups)), max(1, int(desc_host.total_px)))
    sel._host = desc_host
    return sel


# ------------------------------------------------------------------ conv path
def timesblock_forward(x: torch.Tensor, plan: FtnPlan, wblob: torch.Tensor, sel: Selection,
                       norm=None, act_dtype: int = 0, range_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``norm=(gamma, beta, eps)`` appends the model's per-block ``LayerNorm(x + (y - x))``
    (reference :2050-2058) to the same call.  ``range_flag``: see ``new_range_flag``."""
    lib = _lib.load()
    B, L, _ = x.shape
    need = lib.ftn_timesblock_workspace_bytes(C.byref(plan), B, L, sel.max_groups, sel.px_bound)
    if need == 0:
        raise ValueError(f"ftn_timesblock_workspace_bytes rejected the shape (B={B}, L={L})")
    # one workspace per call, from the caching allocator: ordered on the calling stream, private to a graph
    # capture's pool, never shared between calls in flight (a process-wide buffer would be)
    flags = 0
    pre = getattr(sel, "stage_a", None)
    if pre is not None:
        # the selector's finalize launch already ran stage A for THIS x and plan into a workspace it allocated
        ws, x_ptr, plan_addr = pre
        sel.stage_a = None
        if x_ptr != x.data_ptr() or plan_addr != C.addressof(plan) or ws.numel() < need:
            raise RuntimeError("selection carries stage A of a different input or plan")
        flags = 1   # FTN_FWD_STAGE_A_DONE
    else:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    if norm is not None:
        g, b, eps = norm
        check(lib.ftn_timesblock_forward_norm(_ptr(x), _ptr(y), B, L, C.byref(plan), _ptr(wblob), _ptr(sel.desc),
                                              _ptr(sel.weights), sel.max_groups, sel.px_bound, flags, _ptr(g), _ptr(b),
                                              float(eps),
                                              _ptr(ws), ws.numel(), _stream(x.device), _ptr_or_null(range_flag)),
              "ftn_timesblock_forward_norm")
        return y
    check(lib.ftn_timesblock_forward(_ptr(x), _ptr(y), B, L, C.byref(plan), _ptr(wblob), _ptr(sel.desc),
                                     _ptr(sel.weights), sel.max_groups, sel.px_bound, int(act_dtype), flags, _ptr(ws),
                                     ws.numel(), _stream(x.device), _ptr_or_null(range_flag)), "ftn_timesblock_forward")
    return y


def residual_layernorm(x: torch.Tensor, new: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                       eps: float) -> torch.Tensor:
    """LayerNorm over the last axis of ``x + (new - x)`` (fp32, contiguous)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.ftn_residual_layernorm(_ptr(x), _ptr(new), _ptr(out), x.numel() // Cc, Cc, _ptr(gamma), _ptr(beta),
                                     float(eps), _stream(x.device)), "ftn_residual_layernorm")
    return out


# ------------------------------------------------------------------ model shell
def head_forward(hidden: torch.Tensor, w_mu: torch.Tensor, b_mu: torch.Tensor, w_sigma: torch.Tensor,
                 b_sigma: torch.Tensor, tail: torch.Tensor, hist: int, late, floor_vec, floor_scalar: float):
    """Fused rate / dispersion heads.  ``hidden`` [B,S,D] contiguous fp32; ``tail`` a (possibly strided
    along the batch) view [B,hist,N] of the input window; ``late`` None or contiguous [1|B,S,N].
    Returns ``(rate, dispersion, bad_flag)``; ``bad_flag`` is a 1-element int32 device tensor."""
    lib = _lib.load()
    B, S, D = hidden.shape
    N = w_mu.shape[0]
    if tail.stride(2) != 1 or tail.stride(1) != N or tail.shape != (B, hist, N):
        raise ValueError("tail must be a [B, hist, N] view with contiguous rows")
    dev = hidden.device
    rate = torch.empty(B, S, N, dtype=torch.float32, device=dev)
    disp = torch.empty(B, S, N, dtype=torch.float32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    late_bs = 0
    if late is not None:
        if late.shape[-2:] != (S, N) or not late.is_contiguous():
            raise ValueError("late bias must be contiguous [1|B, S, N]")
        late_bs = S * N if late.shape[0] == B and B > 1 else 0
    check(lib.ftn_head_forward(_ptr(hidden), B * S, S, D, N, _ptr(w_mu), _ptr(b_mu), _ptr(w_sigma), _ptr(b_sigma),
                               _ptr(tail), tail.stride(0) if B > 1 else 0, int(hist),
                               _ptr(late) if late is not None else None, late_bs,
                               _ptr(floor_vec) if floor_vec is not None else None, float(floor_scalar),
                               _ptr(rate), _ptr(disp), _ptr(bad), _stream(dev)), "ftn_head_forward")
    return rate, disp, bad


def embed_forward(window: torch.Tensor, weight: torch.Tensor, add, norm=None) -> torch.Tensor:
    """``window`` [B,L,N] fp32 view with contiguous rows; ``weight`` [D,N]; ``add`` None or contiguous
    [1|B,L,D]; ``norm`` None or ``(gamma, beta, eps)``.  Returns [B,L,D]."""
    lib = _lib.load()
    B, L, N = window.shape
    D = weight.shape[0]
    if window.stride(2) != 1 or window.stride(1) != N:
        raise ValueError("window rows must be contiguous")
    add_bs = 0
    if add is not None:
        if add.shape[-2:] != (L, D) or not add.is_contiguous():
            raise ValueError("add must be contiguous [1|B, L, D]")
        add_bs = L * D if add.dim() == 3 and add.shape[0] == B and B > 1 else 0
    out = torch.empty(B, L, D, dtype=torch.float32, device=window.device)
    g, b, eps = norm if norm is not None else (None, None, 0.0)
    check(lib.ftn_embed_forward(_ptr(window), window.stride(0) if B > 1 else 0, B, L, N, _ptr(weight), D,
                                _ptr(add) if add is not None else None, add_bs,
                                _ptr(g) if g is not None else None, _ptr(b) if b is not None else None, float(eps),
                                _ptr(out), _stream(window.device)), "ftn_embed_forward")
    return out


# ------------------------------------------------------------------ LRTC
def lrtc_forward(coeff: torch.Tensor, L: int, scale: torch.Tensor, x: torch.Tensor | None) -> torch.Tensor:
    lib = _lib.load()
    B, N, R = coeff.shape
    st = state(coeff.device)
    basis = st.lrtc_basis(L, R)
    out = torch.empty(B, L, N, dtype=torch.float32, device=coeff.device)
    check(lib.ftn_lrtc_forward(_ptr(coeff), _ptr(basis), _ptr(scale), _ptr(x) if x is not None else None,
                               _ptr(out), B, L, N, R, _stream(coeff.device)), "ftn_lrtc_forward")
    return out


def selftest_mfma(device: torch.device) -> torch.Tensor:
    lib = _lib.load()
    out = torch.zeros(16, 16, dtype=torch.float32, device=device)
    check(lib.ftn_selftest_mfma(_ptr(out), _stream(device)), "ftn_selftest_mfma")
    return out
