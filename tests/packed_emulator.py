"""numpy emulation of what the HIP kernels do with a packed weight blob
(stages A-F of flow-timesnet_amd/csrc/inception.hip).  Test infrastructure: it
lets the CPU suite validate the host-side folding/packing (pack.py) and the stage
algebra (live zero tail pixels, zero conv halo, r = res2 - x) against the oracle
without a GPU."""
import numpy as np
from scipy.special import erf


def _act(v, act):
    return np.maximum(v, 0.0) if act == 1 else 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))


def _unpack_conv(blob, off, kh, kw, cinP, coutP):
    n = kh * kw * cinP * coutP
    v = blob[off:off + n].reshape(kh, kw, cinP // 16, coutP // 16, 4, 16, 4)   # [dy][dx][cc][co][q][j][e]
    w = v.transpose(3, 5, 2, 4, 6, 0, 1).reshape(coutP, cinP, kh, kw)           # [co,j][cc,q,e][dy][dx]
    return w.astype(np.float64)


def _conv_same(img, w, b):
    """img[B][H][W][cin], w[cout][cin][kh][kw] -> [B][H][W][cout], zero padding."""
    B, H, W, cin = img.shape
    cout, _, kh, kw = w.shape
    hy, hx = kh // 2, kw // 2
    pad = np.zeros((B, H + 2 * hy, W + 2 * hx, cin))
    pad[:, hy:hy + H, hx:hx + W] = img
    out = np.zeros((B, H, W, cout)) + b
    for dy in range(kh):
        for dx in range(kw):
            out += pad[:, dy:dy + H, dx:dx + W] @ w[:, :, dy, dx].T
    return out


def _mat(blob, off, rows, cols):
    return blob[off:off + rows * cols].reshape(rows, cols).astype(np.float64)


def _vec(blob, off, n):
    return blob[off:off + n].astype(np.float64)


def emulate(x, blob, plan, group_periods, weights):
    """x[B][L][C] fp32, weights[B][G] -> y[B][L][C] (float64 arithmetic)."""
    B, L, C = x.shape
    CP, FP, act = plan.CP, plan.FP, plan.act
    xp = np.zeros((B, L, CP))
    xp[:, :, :C] = x
    y = x.astype(np.float64).copy()
    for g, p in enumerate(group_periods):
        pad = (-L) % p
        P = L + pad
        cyc = P // p
        u = np.zeros((B, P, CP))
        u[:, :L] = xp                                          # tail pixels are live zeros
        if plan.mode == 0:
            CA = plan.nbr * plan.MP
            MP = plan.MP

            def gconv(a, woffs, boff):
                out = np.zeros((B, P, CA))
                bias = _vec(blob, boff, CA)
                for k in range(plan.nbr):
                    w = _unpack_conv(blob, woffs[k], plan.kh[k], plan.kw[k], MP, MP)
                    img = a[:, :, k * MP:(k + 1) * MP].reshape(B, cyc, p, MP)
                    out[:, :, k * MP:(k + 1) * MP] = _conv_same(img, w, bias[k * MP:(k + 1) * MP]).reshape(B, P, MP)
                return out

            a = u @ _mat(blob, plan.w_in1, CA, CP).T + _vec(blob, plan.b_in1, CA)          # A
            m = gconv(a, plan.w_conv1, plan.b_conv1)                                        # B
            z = m @ _mat(blob, plan.w_out1, FP, CA).T + _vec(blob, plan.b_out1, FP)         # C
            res1 = u @ _mat(blob, plan.w_res1, FP, CP).T + _vec(blob, plan.b_res1, FP) if plan.res1 else u
            gh = _act(_act(z, act) + res1, act)
            n_oa = CA // 16
            rows = CA + (CP if plan.res2 else 0)
            oc = gh @ _mat(blob, plan.w_c2, rows, FP).T + _vec(blob, plan.b_c2, rows)
            a2 = oc[:, :, :CA]
            r = (oc[:, :, CA:] if plan.res2 else gh) - u
            m2 = gconv(a2, plan.w_conv2, plan.b_conv2)                                      # D
            z2 = m2 @ _mat(blob, plan.w_out2, CP, CA).T + _vec(blob, plan.b_out2, CP)       # E
            delta = _act(z2, act) + r
        else:
            kh, kw = plan.kh[0], plan.kw[0]
            w1 = _unpack_conv(blob, plan.w_conv1[0], kh, kw, CP, FP)
            m = _conv_same(u.reshape(B, cyc, p, CP), w1, _vec(blob, plan.b_conv1, FP)).reshape(B, P, FP)
            res1 = u @ _mat(blob, plan.w_res1, FP, CP).T + _vec(blob, plan.b_res1, FP) if plan.res1 else u
            gh = _act(_act(m, act) + res1, act)
            res2 = gh @ _mat(blob, plan.w_res2, CP, FP).T + _vec(blob, plan.b_res2, CP) if plan.res2 else gh
            r = res2 - u
            w2 = _unpack_conv(blob, plan.w_conv2[0], kh, kw, FP, CP)
            m2 = _conv_same(gh.reshape(B, cyc, p, FP), w2, _vec(blob, plan.b_conv2, CP)).reshape(B, P, CP)
            delta = _act(m2, act) + r
        y += weights[:, g].reshape(B, 1, 1) * delta[:, :L, :C]
    return y


# ---- f16x2 engine: replay of the packed fp16 pieces (three per weight fragment) with two-piece activations ----
def _h2_bits(blob, off, n16):
    """n16 fp16 values stored as 16-bit patterns from float offset `off`."""
    return blob[off:off + (n16 + 1) // 2].view(np.uint16)[:n16].view(np.float16).astype(np.float64)


def _unpack_conv_h2(blob, off, kh, kw, cinP, coutP):
    """-> (A1, A2, A3) each [cout][cin][kh*kw]  (pack.py _pack_conv_h2)."""
    nt = kh * kw
    S = (nt + 1) // 2
    n = (cinP // 16) * (coutP // 16) * S * 3 * 4 * 16 * 8
    v = _h2_bits(blob, off, n).reshape(cinP // 16, coutP // 16, S, 3, 2, 2, 16, 8)  # [cc][co][slab][piece][tp][half][i][e]
    v = v.transpose(3, 1, 6, 0, 5, 7, 2, 4)                                          # [piece][co][i][cc][half][e][slab][tp]
    v = v.reshape(3, coutP, cinP, 2 * S)[:, :, :, :nt]
    return v[0], v[1], v[2]


def _split_act(v):
    hi = v.astype(np.float32).astype(np.float16)
    lo = ((v.astype(np.float32) - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def _unpack_cfrag_h2(blob, plan, CA, CP, FP):
    """-> pieces of W_out1~ [FP][KMpad], W_res1~ [FP][CPpad], W_c~ [rows][FP] as lists (A1, A2, A3)."""
    nsKM, nsCP = (CA + 31) // 32, (CP + 31) // 32
    n_ot = CA // 16 + CP // 16
    per = plan.cfragbf_per_chunk
    nch = plan.n_hchunks
    v = _h2_bits(blob, plan.w_cfragbf, nch * per * 3 * 512).reshape(nch, per, 3, 4, 16, 8)   # [hc][frag][piece][qa][i][e]
    Wo = np.zeros((3, nch * 32, nsKM * 32)); Wr = np.zeros((3, nch * 32, nsCP * 32)); Wc = np.zeros((3, n_ot * 16, nch * 32))
    for hc in range(nch):
        k = 0
        for t in range(2):
            for s_ in range(nsKM):
                Wo[:, 32 * hc + 16 * t:32 * hc + 16 * t + 16, 32 * s_:32 * s_ + 32] = v[hc, k].transpose(0, 2, 1, 3).reshape(3, 16, 32); k += 1
        for t in range(2):
            for s_ in range(nsCP):
                Wr[:, 32 * hc + 16 * t:32 * hc + 16 * t + 16, 32 * s_:32 * s_ + 32] = v[hc, k].transpose(0, 2, 1, 3).reshape(3, 16, 32); k += 1
        perm = [32 * hc + (4 * qa + e if e < 4 else 16 + 4 * qa + e - 4) for qa in range(4) for e in range(8)]
        for o in range(n_ot):
            Wc[:, 16 * o:16 * o + 16, perm] = v[hc, k].transpose(0, 2, 1, 3).reshape(3, 16, 32); k += 1
    return Wo[:, :FP, :CA], Wr[:, :FP, :CP], Wc[:, :, :FP]


def _h2_matmul(v, pieces):
    """sum over K of the three f16x2 products (A1 hi + A2 lo' + A3 hi); v[..., K], pieces (3, rows, K)."""
    hi, lo = _split_act(v)
    return hi @ (pieces[0] + pieces[2]).T + lo @ pieces[1].T


def emulate_h2(x, blob, plan, group_periods, weights):
    """The f16x2 engine's arithmetic (bottleneck mode with both res_proj convs): fp32 stage A and E as in
    ``emulate``; conv and stage-C GEMMs from the packed fp16 pieces, prescaled biases and scales of the plan."""
    assert plan.mode == 0 and plan.engine == 3 and plan.res1 and plan.res2
    split_c = plan.cfragbf_per_chunk > 0      # stage C on the fp16 pipe only for the tuned shapes; fp32 MFMA otherwise
    B, L, C = x.shape
    CP, FP, act, MP = plan.CP, plan.FP, plan.act, plan.MP
    CA = plan.nbr * MP
    xp = np.zeros((B, L, CP)); xp[:, :, :C] = x
    y = x.astype(np.float64).copy()
    if split_c:
        Wo, Wr, Wc = _unpack_cfrag_h2(blob, plan, CA, CP, FP)
    for g, p in enumerate(group_periods):
        pad = (-L) % p
        P = L + pad
        cyc = P // p
        u = np.zeros((B, P, CP)); u[:, :L] = xp

        def gconv(a, woffs, boff_s, scs):
            out = np.zeros((B, P, CA))
            bias = _vec(blob, boff_s, CA)
            hi, lo = _split_act(a)
            for k in range(plan.nbr):
                kh, kw = plan.kh[k], plan.kw[k]
                A1, A2, A3 = _unpack_conv_h2(blob, woffs[k], kh, kw, MP, MP)
                sl = slice(k * MP, (k + 1) * MP)
                zero = np.zeros(MP)
                acc = _conv_same(hi[:, :, sl].reshape(B, cyc, p, MP), (A1 + A3).reshape(MP, MP, kh, kw), bias[sl])
                acc += _conv_same(lo[:, :, sl].reshape(B, cyc, p, MP), A2.reshape(MP, MP, kh, kw), zero)
                out[:, :, sl] = acc.reshape(B, P, MP) / scs[k]
            return out

        a = u @ _mat(blob, plan.w_in1, CA, CP).T + _vec(blob, plan.b_in1, CA)               # A (fp32 MFMA)
        m = gconv(a, plan.w_convbf1, plan.b_conv1s, plan.sc_conv1)                           # B
        if split_c:
            z = (_h2_matmul(m, Wo) + _vec(blob, plan.b_out1s, FP)) / plan.sc_out1           # C layer 1
            hacc = _act(z, act) * plan.sc_res1 + _vec(blob, plan.b_res1s, FP) + _h2_matmul(u[:, :, :CP], Wr)
            gh = _act(hacc / plan.sc_res1, act)
            oc = _h2_matmul(gh, Wc) + _vec(blob, plan.b_c2s, CA + CP)
            a2 = oc[:, :, :CA] / plan.sc_a2
            r = oc[:, :, CA:] / plan.sc_r2 - u
        else:
            z = m @ _mat(blob, plan.w_out1, FP, CA).T + _vec(blob, plan.b_out1, FP)
            gh = _act(_act(z, act) + u @ _mat(blob, plan.w_res1, FP, CP).T + _vec(blob, plan.b_res1, FP), act)
            oc = gh @ _mat(blob, plan.w_c2, CA + CP, FP).T + _vec(blob, plan.b_c2, CA + CP)
            a2 = oc[:, :, :CA]
            r = oc[:, :, CA:] - u
        m2 = gconv(a2, plan.w_convbf2, plan.b_conv2s, plan.sc_conv2)                         # D
        z2 = m2 @ _mat(blob, plan.w_out2, CP, CA).T + _vec(blob, plan.b_out2, CP)           # E (fp32 MFMA)
        delta = _act(z2, act) + r
        y += weights[:, g].reshape(B, 1, 1) * delta[:, :L, :C]
    return y
