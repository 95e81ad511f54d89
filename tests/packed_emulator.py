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
