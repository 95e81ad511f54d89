"""Seeded synthetic weights and inputs for the TimesBlock path.

Everything here is generated with ``numpy.random.RandomState`` (a frozen legacy
stream), so the build container, the GPU box and the golden-vector script all
see bit-identical tensors without shipping large fixtures.

Key names follow the reference ``state_dict`` layout of ``TimesBlock.inception``
(``models/timesnet.py:578-590, 622-637, 744-762`` of the reference):
``{0,2}.paths.{j}.branch.{i}.{weight,bias}``, ``{0,2}.proj.*``, ``{0,2}.res_proj.*``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

KernelSet = Sequence[Tuple[int, int]]


def parse_kernel_set(kernel_set) -> List[Tuple[int, int]]:
    """Same accepted spellings as the reference (``models/timesnet.py:609-621``)."""
    out: List[Tuple[int, int]] = []
    for k in kernel_set:
        if isinstance(k, (tuple, list)):
            if len(k) != 2:
                raise ValueError("kernel_set entries must be (kh, kw) pairs")
            kh, kw = k
        else:
            kh = kw = int(k)
        out.append((int(kh), int(kw)))
    if not out:
        raise ValueError("kernel_set must contain at least one kernel size")
    return out


def bottleneck_mid(in_ch: int, out_ch: int, ratio: float):
    """``None`` for the single-conv branch (``models/timesnet.py:575``), else
    ``max(1, ceil(min(in,out)/ratio))`` (``:582-585``)."""
    if math.isclose(ratio, 1.0, rel_tol=1e-9, abs_tol=1e-9):
        return None
    return max(1, int(math.ceil(min(in_ch, out_ch) / float(ratio))))


def _conv_init(rs: np.random.RandomState, out_ch: int, in_ch: int, kh: int, kw: int):
    bound = 1.0 / math.sqrt(in_ch * kh * kw)
    w = rs.uniform(-bound, bound, size=(out_ch, in_ch, kh, kw)).astype(np.float32)
    b = rs.uniform(-bound, bound, size=(out_ch,)).astype(np.float32)
    return w, b


def inception_shapes(d_model: int, d_ff: int, kernel_set: KernelSet, ratio: float):
    """Ordered (key, shape) list of ``TimesBlock.inception``'s parameters."""
    ks = parse_kernel_set(kernel_set)
    shapes = []
    for blk, (cin, cout) in (("0", (d_model, d_ff)), ("2", (d_ff, d_model))):
        mid = bottleneck_mid(cin, cout, ratio)
        for j, (kh, kw) in enumerate(ks):
            if mid is None:
                shapes.append((f"{blk}.paths.{j}.branch.0", (cout, cin, kh, kw)))
            else:
                shapes.append((f"{blk}.paths.{j}.branch.0", (mid, cin, 1, 1)))
                shapes.append((f"{blk}.paths.{j}.branch.1", (mid, mid, kh, kw)))
                shapes.append((f"{blk}.paths.{j}.branch.2", (cout, mid, 1, 1)))
        shapes.append((f"{blk}.proj", (cout, cout * len(ks), 1, 1)))
        if cin != cout:
            shapes.append((f"{blk}.res_proj", (cout, cin, 1, 1)))
    return shapes


def make_inception_params(
    d_model: int, d_ff: int, kernel_set: KernelSet, ratio: float, seed: int = 0
) -> Dict[str, np.ndarray]:
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights and biases, reference key names."""
    rs = np.random.RandomState(seed)
    params: Dict[str, np.ndarray] = {}
    for key, (o, i, kh, kw) in inception_shapes(d_model, d_ff, kernel_set, ratio):
        w, b = _conv_init(rs, o, i, kh, kw)
        params[key + ".weight"] = w
        params[key + ".bias"] = b
    return params


PLANTED = {
    # planted periods / amplitudes; gives top-k gaps >= 0.25.
    "periods_336": (24, 168, 7, 12, 84),
    "periods_720": (24, 168, 144, 12, 7),
    "amps": (2.0, 1.5, 1.0, 0.7, 0.5),
}


def make_input(
    B: int, L: int, C: int, seed: int = 0, planted: Sequence[float] | None = None,
    amps: Sequence[float] | None = None, noise: float = 1.0,
) -> np.ndarray:
    """``noise*randn(B,L,C) + sum_j a_j sin(2*pi*t/p_j + phi_{b,c})`` as fp32."""
    rs = np.random.RandomState(1000 + seed)
    x = noise * rs.standard_normal(size=(B, L, C))
    if planted is None:
        planted = PLANTED["periods_720"] if L >= 720 else PLANTED["periods_336"]
        planted = [p for p in planted if p < L]
    if amps is None:
        amps = PLANTED["amps"]
    t = np.arange(L, dtype=np.float64).reshape(1, L, 1)
    for p, a in zip(planted, amps):
        phi = rs.uniform(0.0, 2.0 * math.pi, size=(B, 1, C))
        x = x + a * np.sin(2.0 * math.pi * t / float(p) + phi)
    return x.astype(np.float32)
