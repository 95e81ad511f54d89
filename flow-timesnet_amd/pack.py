"""Host-side folding and packing of ``TimesBlock.inception`` weights for the HIP
kernels (numpy only; fp64 folding, fp32 result).

Input: the reference ``state_dict`` of ``nn.Sequential(InceptionBlock, act,
InceptionBlock)`` (reference ``models/timesnet.py:744-762``; key names
``{0,2}.paths.{j}.branch.{i}.*``, ``{0,2}.proj.*``, ``{0,2}.res_proj.*``).

Algebra (exact, SURVEY finding 5): every ``InceptionBranch`` is linear, so
``proj(cat_k branch_k(u))`` = ``sum_k P_k branch_k(u) + b_proj`` with ``P_k`` the
k-th column block of ``proj.weight``:

* bottleneck branch (``1x1 -> kxk -> 1x1``, :586-590): the last 1x1 folds into
  ``P_k``:  ``W_out[:, k] = P_k W3_k``, ``b_out = b_proj + sum_k P_k b3_k``.
  The first 1x1 can NOT be folded through the kxk conv (its bias is absent in
  the zero-padded halo), so it stays a separate stage.
* single-conv branch (ratio 1, :578-580): ``P_k`` folds into the conv itself and
  the per-kernel convs merge into one conv of the largest kernel size (smaller
  kernels are centred, which preserves 'same' zero padding).

Packed conv weights are lane-linear for ``v_mfma_f32_16x16x4_f32``:
``[tap][cin/16][cout/16][q][j][e] = W[16*co + j][16*cc + 4*q + e][dy][dx]``
(one contiguous KiB per wave-wide A-fragment load).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np

from . import synth
from .lib import FTN_MAXBR, FtnPlan


def _pad16(v: int) -> int:
    return (int(v) + 15) // 16 * 16


class _Blob:
    def __init__(self) -> None:
        self.parts = []
        self.n = 0

    def add(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self.n
        pad = (-arr.size) % 64
        self.parts.append(arr)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.n += arr.size + pad
        return off

    def finish(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(0, np.float32)


def _pack_conv(w: np.ndarray, cinP: int, coutP: int) -> np.ndarray:
    """w[cout][cin][kh][kw] -> [tap][cc][co][q][j][e] (zero padded)."""
    cout, cin, kh, kw = w.shape
    wp = np.zeros((coutP, cinP, kh, kw), np.float64)
    wp[:cout, :cin] = w
    # [co][j][cc][q][e][dy][dx] -> [dy][dx][cc][co][q][j][e]
    v = wp.reshape(coutP // 16, 16, cinP // 16, 4, 4, kh, kw)
    v = v.transpose(5, 6, 2, 0, 3, 1, 4)
    return np.ascontiguousarray(v).astype(np.float32)


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even), returned as fp32 values with 16 low bits clear."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    """hi + mid + lo bf16 pieces of fp32 data (what csrc/ftn_common.h store_p3 does)."""
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    l = bf16_rne((r1 - m).astype(np.float32))
    return h, m, l


def _bf16_bits_as_f32(pieces: np.ndarray) -> np.ndarray:
    """bf16 values (held as fp32) -> their 16-bit patterns packed two per float32 word."""
    bits = (np.ascontiguousarray(pieces, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
    flat = bits.reshape(-1)
    if flat.size % 2:
        flat = np.concatenate([flat, np.zeros(1, np.uint16)])
    return flat.view(np.float32)


def _pack_conv_bf(w: np.ndarray, cinP: int, coutP: int) -> np.ndarray:
    """w[cout][cin][kh][kw] -> bf16x3 K=32 fragments [cc][co][slab][piece][lane][8]: lane
    (i = lane & 15, qa = lane >> 4) holds W[16co + i][16cc + 8(qa & 1) + e][tap 2*slab + (qa >> 1)]."""
    cout, cin, kh, kw = w.shape
    nt = kh * kw
    S = (nt + 1) // 2
    wp = np.zeros((coutP, cinP, 2 * S), np.float32)
    wp[:cout, :cin, :nt] = w.reshape(cout, cin, nt)
    v = wp.reshape(coutP // 16, 16, cinP // 16, 2, 8, S, 2)          # [co][i][cc][half][e][slab][tp]
    v = v.transpose(2, 0, 5, 6, 3, 1, 4)                                # [cc][co][slab][tp][half][i][e]
    v = np.ascontiguousarray(v).reshape(cinP // 16, coutP // 16, S, 4, 16, 8)   # qa = 2*tp + half
    h, m, l = split3(v)
    out = np.stack([h, m, l], axis=3)                                   # [cc][co][slab][piece][qa][i][e]
    return _bf16_bits_as_f32(out)


# ---- f16x2 engine: three fp16 weight pieces of the power-of-two-prescaled matrix (csrc/ftn_common.h) ----
H2_LOSHIFT = 11          # activations: lo' = (x - fp16(x)) * 2^11; weights: A2 = fp16(A1 * 2^-11)


def pow2_scale(W: np.ndarray) -> float:
    """Power of two ``sc`` with ``max|sc W|`` in [2^10, 2^11): the three fp16 pieces of ``sc W`` - including
    ``A2 = A1 2^-11`` and the remainder ``A3 ~ 2^-11 |sc W|`` - then stay fp16 normals for every weight that is
    not negligible beside the largest one, and fp16 cannot overflow (max 65504)."""
    m = float(np.max(np.abs(W))) if W.size else 0.0
    if not np.isfinite(m) or m <= 0.0:
        return 1.0
    return float(2.0 ** (10 - int(np.floor(np.log2(m)))))


def split_h2_weights(Ws: np.ndarray):
    """(A1, A2, A3) fp16 pieces (held as fp32) of an already prescaled weight array."""
    Ws = np.asarray(Ws, dtype=np.float32)
    a1 = Ws.astype(np.float16)
    a2 = (a1.astype(np.float32) * np.float32(2.0 ** -H2_LOSHIFT)).astype(np.float16)
    a3 = (Ws - a1.astype(np.float32)).astype(np.float16)
    return a1, a2, a3


def split_h2_act(x: np.ndarray):
    """(hi, lo') fp16 pieces of activations, as csrc/ftn_common.h split_h2 (round to nearest even)."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2.0 ** H2_LOSHIFT)).astype(np.float16)
    return hi, lo


def _f16_bits_as_f32(pieces: np.ndarray) -> np.ndarray:
    """fp16 array -> its 16-bit patterns packed two per float32 word."""
    flat = np.ascontiguousarray(pieces, dtype=np.float16).view(np.uint16).reshape(-1)
    if flat.size % 2:
        flat = np.concatenate([flat, np.zeros(1, np.uint16)])
    return flat.view(np.float32)


def _pack_conv_h2(w: np.ndarray, cinP: int, coutP: int, sc: float) -> np.ndarray:
    """As ``_pack_conv_bf`` with the f16x2 pieces of ``sc * w``: [cc][co][slab][piece][lane][8]."""
    cout, cin, kh, kw = w.shape
    nt = kh * kw
    S = (nt + 1) // 2
    wp = np.zeros((coutP, cinP, 2 * S), np.float32)
    wp[:cout, :cin, :nt] = (np.asarray(w, np.float64) * sc).reshape(cout, cin, nt)
    v = wp.reshape(coutP // 16, 16, cinP // 16, 2, 8, S, 2)          # [co][i][cc][half][e][slab][tp]
    v = v.transpose(2, 0, 5, 6, 3, 1, 4)                                # [cc][co][slab][tp][half][i][e]
    v = np.ascontiguousarray(v).reshape(cinP // 16, coutP // 16, S, 4, 16, 8)   # qa = 2*tp + half
    a1, a2, a3 = split_h2_weights(v)
    return _f16_bits_as_f32(np.stack([a1, a2, a3], axis=3))           # [cc][co][slab][piece][qa][i][e]


def _frag_h2(W, R: int, cols) -> np.ndarray:
    """K=32 fp16 A fragment of the (prescaled) matrix W, three pieces: [piece][qa][i][e] (cf. ``_frag_bf``)."""
    blk = np.zeros((16, 32), np.float32)
    if W is not None:
        r0 = 16 * R
        rows = min(16, max(0, W.shape[0] - r0))
        for k, c in enumerate(cols):
            if c >= 0 and c < W.shape[1] and rows > 0:
                blk[:rows, k] = W[r0:r0 + rows, c]
    lanes = blk.reshape(16, 4, 8).transpose(1, 0, 2)              # [qa][i][e]
    return np.stack(split_h2_weights(lanes), axis=0)


def _pack_cfrag_h2(W_out, W_res, W_c, FP: int, nsKM: int, nsCP: int, n_ot: int) -> np.ndarray:
    """``_pack_cfrag_bf`` for the f16x2 engine; the matrices arrive prescaled."""
    nch = (FP + 31) // 32
    per = 2 * nsKM + 2 * nsCP + n_ot
    out = np.zeros((nch, max(per, 1), 3, 4, 16, 8), np.float16)
    for hc in range(nch):
        k = 0
        for t in range(2):
            for s_ in range(nsKM):
                out[hc, k] = _frag_h2(W_out, hc * 2 + t, list(range(32 * s_, 32 * s_ + 32))); k += 1
        for t in range(2):
            for s_ in range(nsCP):
                out[hc, k] = _frag_h2(W_res, hc * 2 + t, list(range(32 * s_, 32 * s_ + 32))); k += 1
        perm = [32 * hc + (4 * qa + e if e < 4 else 16 + 4 * qa + e - 4) for qa in range(4) for e in range(8)]
        for o in range(n_ot):
            out[hc, k] = _frag_h2(W_c, o, perm); k += 1
    return _f16_bits_as_f32(out)


def _frag(W: np.ndarray, R: int, S: int) -> np.ndarray:
    """16x16 block (rows 16R.., cols 16S..) of W as a lane-linear MFMA A fragment:
    [lane = 16*q + j][e] = W[16R + j][16S + 4q + e]; zero outside W."""
    blk = np.zeros((16, 16))
    r0, c0 = 16 * R, 16 * S
    if r0 < W.shape[0] and c0 < W.shape[1]:
        sub = W[r0:r0 + 16, c0:c0 + 16]
        blk[:sub.shape[0], :sub.shape[1]] = sub
    return blk.reshape(16, 4, 4).transpose(1, 0, 2).reshape(-1)


CHUNK_TILES = 2   # hidden row tiles (of 16 channels) per stage-C chunk; must match HT in csrc/inception.hip


def _pack_cfrag(W_out, W_res, W_c, FP: int, nKM: int, nCP: int, n_ot: int) -> np.ndarray:
    """Stage-C fragments grouped per 32-channel hidden chunk (see FtnPlan.w_cfrag)."""
    HT = CHUNK_TILES
    nch = (FP + 16 * HT - 1) // (16 * HT)
    per = HT * (nKM + nCP + n_ot)
    out = np.zeros((nch, max(per, 1), 256))
    for hc in range(nch):
        k = 0
        for t in range(HT):
            for s_ in range(nKM):
                out[hc, k] = _frag(W_out, hc * HT + t, s_); k += 1
        for t in range(HT):
            for s_ in range(nCP):
                out[hc, k] = _frag(W_res, hc * HT + t, s_); k += 1
        for t in range(HT):
            for o in range(n_ot):
                out[hc, k] = _frag(W_c, o, hc * HT + t); k += 1
    return out


def _frag_bf(W, R: int, cols) -> np.ndarray:
    """K=32 bf16 A fragment, three pieces: [piece][lane = 16*qa + i][e] = W[16R + i][cols[8*qa + e]]
    (cols: 32 column indices, -1 = zero)."""
    blk = np.zeros((16, 32), np.float32)
    if W is not None:
        r0 = 16 * R
        rows = min(16, max(0, W.shape[0] - r0))
        for k, c in enumerate(cols):
            if c >= 0 and c < W.shape[1] and rows > 0:
                blk[:rows, k] = W[r0:r0 + rows, c]
    lanes = blk.reshape(16, 4, 8).transpose(1, 0, 2)              # [qa][i][e]
    h, m, l = split3(lanes)
    return np.stack([h, m, l], axis=0)                               # [piece][qa][i][e]


def _pack_cfrag_bf(W_out, W_res, W_c, FP: int, nsKM: int, nsCP: int, n_ot: int) -> np.ndarray:
    """Stage-C weights for the bf16x3 engine, per 32-channel hidden chunk:
    [ W_out1: 2 tiles x nsKM slabs | W_res1: 2 tiles x nsCP slabs | W_c: n_ot tiles ] x 3 pieces,
    1 KiB per piece.  Layer-2 columns follow the accumulator order of the hidden tile pair:
    k = 8*qa + e -> hidden channel 32*hc + (e < 4 ? 4*qa + e : 16 + 4*qa + e - 4)."""
    nch = (FP + 31) // 32
    per = 2 * nsKM + 2 * nsCP + n_ot
    out = np.zeros((nch, max(per, 1), 3, 4, 16, 8), np.float32)
    for hc in range(nch):
        k = 0
        for t in range(2):
            for s_ in range(nsKM):
                out[hc, k] = _frag_bf(W_out, hc * 2 + t, list(range(32 * s_, 32 * s_ + 32))); k += 1
        for t in range(2):
            for s_ in range(nsCP):
                out[hc, k] = _frag_bf(W_res, hc * 2 + t, list(range(32 * s_, 32 * s_ + 32))); k += 1
        perm = [32 * hc + (4 * qa + e if e < 4 else 16 + 4 * qa + e - 4) for qa in range(4) for e in range(8)]
        for o in range(n_ot):
            out[hc, k] = _frag_bf(W_c, o, perm); k += 1
    return _bf16_bits_as_f32(out)


def _check_odd(ks):
    for kh, kw in ks:
        if kh % 2 == 0 or kw % 2 == 0 or kh < 1 or kw < 1:
            raise ValueError(f"kernel sizes must be odd and positive for 'same' padding, got {(kh, kw)}")


ENGINES = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x2": 3}
DEFAULT_ENGINE = "f16x2"


def default_engine() -> str:
    """Conv arithmetic: ``f32`` exact fp32 MFMA; ``f16x2`` (default) two fp16 pieces per activation and three
    per (prescaled) weight, three products per fp32 multiply on the fp16 matrix pipe - fp32-equivalent accuracy
    (2^-22 per operand) for activations below the fp16 maximum 65504, the range the reference's own fp16-autocast
    GPU path assumes; ``bf16x3`` three bf16 pieces, six products, the full fp32 exponent range; ``bf16`` plain
    bf16 operands with fp32 accumulation (BASELINE configs[2]).  Env ``FLOWTIMES_ENGINE``."""
    import os

    e = os.environ.get("FLOWTIMES_ENGINE", DEFAULT_ENGINE).strip().lower()
    if e not in ENGINES:
        raise ValueError(f"FLOWTIMES_ENGINE must be one of {sorted(ENGINES)}, got {e!r}")
    return e


def pack_inception(
    sd: Dict[str, np.ndarray], d_model: int, d_ff: int, kernel_set: Sequence[Tuple[int, int]],
    ratio: float, act: str, engine: str | None = None,
) -> Tuple[np.ndarray, FtnPlan]:
    """Fold + pack the reference ``state_dict`` of ``TimesBlock.inception`` through the C ABI
    (``ftn_inception_pack_weights``, csrc/pack.hip): returns (fp32 weight blob, FtnPlan with float offsets).
    This wrapper only marshals the tensors into ``FtnInceptionBlockWeights``."""
    import ctypes as C

    from . import lib as _lib
    from .lib import FtnInceptionBlockWeights

    ks = synth.parse_kernel_set(kernel_set)
    _check_odd(ks)
    if len(ks) > FTN_MAXBR:
        raise ValueError(f"at most {FTN_MAXBR} kernels are supported, got {len(ks)}")
    eng = ENGINES[engine if engine is not None else default_engine()]
    lib = _lib.load()
    keep = []                                   # the fp32 arrays the C struct points into

    def ptr(key):
        if key not in sd:
            return None
        a = np.ascontiguousarray(np.asarray(sd[key], dtype=np.float32))
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    mid = synth.bottleneck_mid(int(d_model), int(d_ff), ratio)
    blocks = []
    for blk, (cin, cout) in (("0", (d_model, d_ff)), ("2", (d_ff, d_model))):
        w = FtnInceptionBlockWeights()
        for j, (kh, kw) in enumerate(ks):
            for i in range(3 if mid is not None else 1):
                key = f"{blk}.paths.{j}.branch.{i}"
                if f"{key}.weight" not in sd:
                    raise ValueError(f"{key}.weight missing")
                shape = tuple(np.asarray(sd[f"{key}.weight"]).shape)
                want = ((mid, cin, 1, 1), (mid, mid, kh, kw), (cout, mid, 1, 1))[i] if mid is not None else (cout, cin, kh, kw)
                if shape != want:
                    raise ValueError(f"unexpected shape {shape} of {key}.weight, expected {want}")
                w.branch_w[j][i] = ptr(f"{key}.weight")
                w.branch_b[j][i] = ptr(f"{key}.bias")
        w.proj_w, w.proj_b = ptr(f"{blk}.proj.weight"), ptr(f"{blk}.proj.bias")
        w.res_w, w.res_b = ptr(f"{blk}.res_proj.weight"), ptr(f"{blk}.res_proj.bias")
        if w.res_w is None and cin != cout:
            raise ValueError(f"{blk}.res_proj.weight missing although in_ch != out_ch")
        blocks.append(w)
    kh = (C.c_int * len(ks))(*[k[0] for k in ks])
    kw = (C.c_int * len(ks))(*[k[1] for k in ks])
    n = lib.ftn_inception_pack_floats(int(d_model), int(d_ff), len(ks), kh, kw, float(ratio), eng)
    if n == 0:
        _lib.check(-1, "ftn_inception_pack_floats")
    blob = np.empty(n, dtype=np.float32)
    plan = FtnPlan()
    _lib.check(lib.ftn_inception_pack_weights(C.byref(blocks[0]), C.byref(blocks[1]), int(d_model), int(d_ff), len(ks),
                                              kh, kw, float(ratio), 1 if act.lower() == "relu" else 0, eng,
                                              blob.ctypes.data_as(C.c_void_p), n, C.byref(plan)),
               "ftn_inception_pack_weights")
    return blob, plan


def pack_inception_numpy(
    sd: Dict[str, np.ndarray], d_model: int, d_ff: int, kernel_set: Sequence[Tuple[int, int]],
    ratio: float, act: str, engine: str | None = None,
) -> Tuple[np.ndarray, FtnPlan]:
    """The same folding + packing in numpy (round 1's packer): kept as an independent restatement that
    ``tests/test_host_logic.py`` compares the C packer against; the product calls ``pack_inception``."""
    ks = synth.parse_kernel_set(kernel_set)
    _check_odd(ks)
    if len(ks) > FTN_MAXBR:
        raise ValueError(f"at most {FTN_MAXBR} kernels are supported, got {len(ks)}")
    sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
    C, F = int(d_model), int(d_ff)
    CP, FP = _pad16(C), _pad16(F)
    plan = FtnPlan()
    plan.engine = ENGINES[engine if engine is not None else default_engine()]
    plan.C, plan.CP, plan.F, plan.FP = C, CP, F, FP
    plan.act = 1 if act.lower() == "relu" else 0
    mid = synth.bottleneck_mid(C, F, ratio)
    blob = _Blob()
    nk = len(ks)
    for j in range(FTN_MAXBR):
        plan.sc_conv1[j] = plan.sc_conv2[j] = 1.0
    plan.sc_out1 = plan.sc_res1 = plan.sc_a2 = plan.sc_r2 = plan.sc_out2 = 1.0

    def res(blk, cin, cout, cinP, coutP):
        key = f"{blk}.res_proj.weight"
        if key not in sd:
            if cin != cout:
                raise ValueError(f"{key} missing although in_ch != out_ch")
            return 0, 0, 0
        W = np.zeros((coutP, cinP))
        W[:cout, :cin] = sd[key][:, :, 0, 0]
        b = np.zeros(coutP)
        b[:cout] = sd[f"{blk}.res_proj.bias"]
        return 1, blob.add(W), blob.add(b)

    if mid is not None:
        plan.mode = 0
        MP = _pad16(mid)
        plan.MP, plan.nbr = MP, nk
        for j, (kh, kw) in enumerate(ks):
            plan.kh[j], plan.kw[j] = kh, kw
        CA = nk * MP

        convs_bf = []
        h2 = plan.engine == ENGINES["f16x2"]
        conv_sc = []

        def block(blk, cin, cout, cinP, coutP):
            W_in = np.zeros((CA, cinP)); b_in = np.zeros(CA)
            b_conv = np.zeros(CA)
            W_out = np.zeros((coutP, CA)); b_out = np.zeros(coutP)
            b_out[:cout] = sd[f"{blk}.proj.bias"]
            proj = sd[f"{blk}.proj.weight"][:, :, 0, 0]                  # [cout][nk*cout]
            convs = []
            for j in range(nk):
                w1 = sd[f"{blk}.paths.{j}.branch.0.weight"][:, :, 0, 0]  # [mid][cin]
                w2 = sd[f"{blk}.paths.{j}.branch.1.weight"]              # [mid][mid][kh][kw]
                w3 = sd[f"{blk}.paths.{j}.branch.2.weight"][:, :, 0, 0]  # [cout][mid]
                if w1.shape != (mid, cin) or w3.shape != (cout, mid) or w2.shape[:2] != (mid, mid):
                    raise ValueError(f"unexpected bottleneck shapes in {blk}.paths.{j}")
                W_in[j * MP: j * MP + mid, :cin] = w1
                b_in[j * MP: j * MP + mid] = sd[f"{blk}.paths.{j}.branch.0.bias"]
                b_conv[j * MP: j * MP + mid] = sd[f"{blk}.paths.{j}.branch.1.bias"]
                Pk = proj[:, j * cout:(j + 1) * cout]
                W_out[:cout, j * MP: j * MP + mid] = Pk @ w3
                b_out[:cout] += Pk @ sd[f"{blk}.paths.{j}.branch.2.bias"]
                convs.append(_pack_conv(w2, MP, MP))
                sc = pow2_scale(w2) if h2 else 1.0
                conv_sc.append(sc)
                convs_bf.append(_pack_conv_h2(w2, MP, MP, sc) if h2 else _pack_conv_bf(w2, MP, MP))
            return W_in, b_in, convs, b_conv, W_out, b_out

        W_in1, b_in1, convs1, b_conv1, W_out1, b_out1 = block("0", C, F, CP, FP)
        W_in2, b_in2, convs2, b_conv2, W_out2, b_out2 = block("2", F, C, FP, CP)
        plan.w_in1, plan.b_in1 = blob.add(W_in1), blob.add(b_in1)
        for j in range(nk):
            plan.w_conv1[j] = blob.add(convs1[j])
        plan.b_conv1 = blob.add(b_conv1)
        plan.w_out1, plan.b_out1 = blob.add(W_out1), blob.add(b_out1)
        plan.res1, plan.w_res1, plan.b_res1 = res("0", C, F, CP, FP)
        plan.w_in2, plan.b_in2 = blob.add(W_in2), blob.add(b_in2)
        for j in range(nk):
            plan.w_conv2[j] = blob.add(convs2[j])
        plan.b_conv2 = blob.add(b_conv2)
        plan.w_out2, plan.b_out2 = blob.add(W_out2), blob.add(b_out2)
        # res2 + stacked stage-C projection [W_in2 ; W_res2]
        key = "2.res_proj.weight"
        if key in sd:
            Wr2 = np.zeros((CP, FP)); Wr2[:C, :F] = sd[key][:, :, 0, 0]
            br2 = np.zeros(CP); br2[:C] = sd["2.res_proj.bias"]
            plan.res2 = 1
            plan.w_res2, plan.b_res2 = blob.add(Wr2), blob.add(br2)
            plan.w_c2 = blob.add(np.concatenate([W_in2, Wr2], 0))
            plan.b_c2 = blob.add(np.concatenate([b_in2, br2], 0))
        else:
            if C != F:
                raise ValueError("2.res_proj missing although d_ff != d_model")
            plan.res2 = 0
            plan.w_c2, plan.b_c2 = blob.add(W_in2), blob.add(b_in2)
        Wc = np.concatenate([W_in2, Wr2], 0) if plan.res2 else W_in2
        Wr1 = None
        if plan.res1:
            Wr1 = np.zeros((FP, CP)); Wr1[:F, :C] = sd["0.res_proj.weight"][:, :, 0, 0]
        nKM, nCP, n_ot = CA // 16, (CP // 16 if plan.res1 else 0), Wc.shape[0] // 16
        per = CHUNK_TILES * (nKM + nCP + n_ot)
        plan.n_hchunks = (FP + 16 * CHUNK_TILES - 1) // (16 * CHUNK_TILES)
        if n_ot > 16 or per * 1024 * 2 > 160 * 1024:
            # beyond the fused stage-C kernels (csrc/inception.hip stagec_generic): the chain runs as generic
            # pointwise launches on the row-major matrices, no fragments needed
            plan.w_cfrag, plan.cfrag_per_chunk = 0, 0
        else:
            cf = _pack_cfrag(W_out1, Wr1, Wc, FP, nKM, nCP, n_ot)
            plan.w_cfrag = blob.add(cf)
            plan.cfrag_per_chunk = per
        for j in range(nk):
            plan.w_convbf1[j] = blob.add(convs_bf[j])
            plan.w_convbf2[j] = blob.add(convs_bf[nk + j])
        for j in range(FTN_MAXBR):
            plan.sc_conv1[j] = plan.sc_conv2[j] = 1.0
        plan.sc_out1 = plan.sc_res1 = plan.sc_a2 = plan.sc_r2 = plan.sc_out2 = 1.0
        if h2:
            # biases prescaled like their weight matrices: the accumulators start from them (flowtimes.h FtnPlan)
            s1 = np.repeat(np.array(conv_sc[:nk]), MP)
            s2 = np.repeat(np.array(conv_sc[nk:]), MP)
            for j in range(nk):
                plan.sc_conv1[j], plan.sc_conv2[j] = conv_sc[j], conv_sc[nk + j]
            plan.b_conv1s, plan.b_conv2s = blob.add(b_conv1 * s1), blob.add(b_conv2 * s2)
        tuned = (32 < CA <= 64 and 32 < CP <= 64 and n_ot <= 8) or (CA == 96 and CP == 128 and n_ot == 14)
        if plan.res1 and plan.res2 and tuned:         # shapes the split-engine stage-C kernels exist for
            nsKM, nsCP = (CA + 31) // 32, (CP + 31) // 32
            if h2:
                plan.sc_out1, plan.sc_res1 = pow2_scale(W_out1), pow2_scale(Wr1)
                plan.sc_a2, plan.sc_r2 = pow2_scale(W_in2), pow2_scale(Wr2)
                Wcs = np.concatenate([W_in2 * plan.sc_a2, Wr2 * plan.sc_r2], 0)
                plan.b_out1s, plan.b_res1s = blob.add(b_out1 * plan.sc_out1), blob.add(np.concatenate(
                    [sd["0.res_proj.bias"], np.zeros(FP - F)]) * plan.sc_res1)
                plan.b_c2s = blob.add(np.concatenate([b_in2 * plan.sc_a2, br2 * plan.sc_r2], 0))
                plan.w_cfragbf = blob.add(_pack_cfrag_h2((W_out1 * plan.sc_out1).astype(np.float32),
                                                         (Wr1 * plan.sc_res1).astype(np.float32),
                                                         Wcs.astype(np.float32), FP, nsKM, nsCP, n_ot))
            else:
                plan.w_cfragbf = blob.add(_pack_cfrag_bf(W_out1.astype(np.float32), Wr1.astype(np.float32),
                                                         Wc.astype(np.float32), FP, nsKM, nsCP, n_ot))
            plan.cfragbf_per_chunk = 2 * nsKM + 2 * nsCP + n_ot
            # stage E (k_out_h): w_out2 [CP][CA] as K=32 fragments, row tile major
            frag, bits, Wo2 = _frag_bf, _bf16_bits_as_f32, W_out2
            if h2:
                plan.sc_out2 = pow2_scale(W_out2)
                plan.b_out2s = blob.add(b_out2 * plan.sc_out2)
                frag, bits, Wo2 = _frag_h2, _f16_bits_as_f32, W_out2 * plan.sc_out2
            Wo2 = Wo2.astype(np.float32)
            ofb = np.stack([frag(Wo2, o, list(range(32 * s_, 32 * s_ + 32)))
                            for o in range(CP // 16) for s_ in range(nsKM)], 0)
            plan.w_out2fb = blob.add(bits(ofb if h2 else ofb.astype(np.float32)))
    else:
        plan.mode = 1
        plan.MP, plan.nbr = 0, 1
        KH = max(k[0] for k in ks)
        KW = max(k[1] for k in ks)
        plan.kh[0], plan.kw[0] = KH, KW

        def block(blk, cin, cout, cinP, coutP):
            proj = sd[f"{blk}.proj.weight"][:, :, 0, 0]
            Wm = np.zeros((cout, cin, KH, KW))
            bm = sd[f"{blk}.proj.bias"].copy()
            for j, (kh, kw) in enumerate(ks):
                w = sd[f"{blk}.paths.{j}.branch.0.weight"]               # [cout][cin][kh][kw]
                if w.shape != (cout, cin, kh, kw):
                    raise ValueError(f"unexpected conv shape in {blk}.paths.{j}")
                Pk = proj[:, j * cout:(j + 1) * cout]
                oy, ox = (KH - kh) // 2, (KW - kw) // 2
                Wm[:, :, oy:oy + kh, ox:ox + kw] += np.einsum("op,pikl->oikl", Pk, w)
                bm += Pk @ sd[f"{blk}.paths.{j}.branch.0.bias"]
            b = np.zeros(coutP); b[:cout] = bm
            return _pack_conv(Wm, cinP, coutP), b

        c1, bc1 = block("0", C, F, CP, FP)
        c2, bc2 = block("2", F, C, FP, CP)
        plan.w_conv1[0], plan.b_conv1 = blob.add(c1), blob.add(bc1)
        plan.res1, plan.w_res1, plan.b_res1 = res("0", C, F, CP, FP)
        plan.w_conv2[0], plan.b_conv2 = blob.add(c2), blob.add(bc2)
        plan.res2, plan.w_res2, plan.b_res2 = res("2", F, C, FP, CP)
        Wr1 = Wr2 = None
        if plan.res1:
            Wr1 = np.zeros((FP, CP)); Wr1[:F, :C] = sd["0.res_proj.weight"][:, :, 0, 0]
        if plan.res2:
            Wr2 = np.zeros((CP, FP)); Wr2[:C, :F] = sd["2.res_proj.weight"][:, :, 0, 0]
        nCP, n_ot = (CP // 16 if plan.res1 else 0), (CP // 16 if plan.res2 else 0)
        cf = _pack_cfrag(None, Wr1, Wr2, FP, 0, nCP, n_ot)
        plan.w_cfrag = blob.add(cf)
        plan.cfrag_per_chunk, plan.n_hchunks = CHUNK_TILES * (nCP + n_ot), cf.shape[0]
    out = blob.finish()
    plan.total_floats = out.size
    return out, plan


def macs_per_pixel(d_model: int, d_ff: int, kernel_set, ratio: float, folded: bool = False) -> int:
    """MAC/px of the two InceptionBlocks (SURVEY §8a FLOP model); ``folded`` counts
    what the kernels execute after folding proj∘branch[-1]."""
    ks = synth.parse_kernel_set(kernel_set)
    nk = len(ks)
    total = 0
    for cin, cout in ((d_model, d_ff), (d_ff, d_model)):
        mid = synth.bottleneck_mid(cin, cout, ratio)
        if mid is None:
            if folded:
                KH = max(k[0] for k in ks); KW = max(k[1] for k in ks)
                total += cin * cout * KH * KW
            else:
                total += sum(cin * cout * kh * kw for kh, kw in ks) + nk * cout * cout
        else:
            if folded:
                total += sum(cin * mid + mid * mid * kh * kw for kh, kw in ks) + nk * mid * cout
            else:
                total += sum(cin * mid + mid * mid * kh * kw + mid * cout for kh, kw in ks) + nk * cout * cout
        if cin != cout:
            total += cin * cout
    return int(total)
