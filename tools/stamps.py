#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timeline of k_conv / k_mlp from s_memtime stamps
(ftn_debug_stamps).  Usage on the GPU box: python tools/stamps.py [conv|mlp]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge

which = sys.argv[1] if len(sys.argv) > 1 else "conv"
pkg = ge.load_package()
lib = pkg.lib.load()
T = pkg.models.timesnet
dev = torch.device("cuda:0")
B, L, C, K = int(os.environ.get('STAMP_B', '256')), 336, 64, 5
ks = [(3, 3), (5, 5), (7, 7)]
params = pkg.synth.make_inception_params(C, 4 * C, ks, 4.0, seed=0)
blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
blk.period_selector = T.FFTPeriodSelector(K, L)
blk = blk.eval().to(dev)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
with torch.inference_mode():
    for _ in range(3):
        blk(x)
    torch.cuda.synchronize()
    nwg = 2_000_000
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.ftn_debug_stamps(buf.data_ptr(), buf.numel(), (5 if len(sys.argv) > 2 and sys.argv[2] == 'B' else 1) if which == 'conv' else 2)
    blk(x)
    torch.cuda.synchronize()
    lib.ftn_debug_stamps(None, 0, 0)
s = buf.cpu().numpy().reshape(-1, 8)
# both conv launches and the mlp launch write the same buffer: conv D overwrites conv B; mlp uses blockIdx.x ids
allw = s[s[:, 6] > 0] if (s[:, 6] > 0).any() else s[s[:, 0] > 0]
print("dispatched workgroups", len(allw))
live = s[:, 3] > 0
s = s[live].astype(np.float64)
t0 = s[:, 0].min()
print("live workgroups", len(s))
if which == "conv":
    xcc = np.zeros(len(s), dtype=int)
    for xid in sorted(set(xcc)):
        m = xcc == xid
        print(f"  xcc {xid}: n={m.sum():4d} start {(s[m,6].min()-s[:,6].min())/100:7.1f} us  end {(s[m,7].max()-s[:,6].min())/100:7.1f} us")
    st = (s[:, 6] - s[:, 6].min()) / 100.0
    en = (s[:, 7] - s[:, 6].min()) / 100.0
    edges = np.linspace(0, en.max(), 13)
    print("  active WGs at t:", [int(((st <= e) & (en > e)).sum()) for e in edges[:-1]])
if which == "mlp" and (s[:, 7] > 0).all():
    for name, a, b in (("chunk 1 work", 2, 7), ("chunk 1 closing barrier", 7, 4), ("chunk 2 weight refill + barrier", 4, 6)):
        dt = (s[:, b] - s[:, a])
        print(f"{name:26s} mean {dt.mean():9.0f} cyc  p50 {np.median(dt):9.0f}  p90 {np.percentile(dt, 90):9.0f}")
for name, a, b in (("stage", 0, 1), ("compute", 1, 2), ("store", 2, 3), ("total", 0, 3)):
    dt = s[:, b] - s[:, a]
    print(f"{name:8s} mean {dt.mean():10.0f} p50 {np.median(dt):10.0f} p90 {np.percentile(dt, 90):10.0f} max {dt.max():10.0f}")
if which == "conv":
    for taps in sorted(set(s[:, 4].astype(int))):
        m = s[:, 4] == taps
        print(f"taps {taps:3d}: n={m.sum():5d} stage {np.mean(s[m,1]-s[m,0]):8.0f} compute {np.mean(s[m,2]-s[m,1]):8.0f} "
              f"total {np.mean(s[m,3]-s[m,0]):8.0f}  start-span {s[m,0].min()-t0:9.0f}..{s[m,0].max()-t0:9.0f}")
if which == "conv":
    tot = s[:, 3] - s[:, 0]
    order = np.argsort(-tot)[:12]
    print("slowest workgroups: (index in live list, taps, start offset us, total cycles, first-row stage cycles)")
    for i in order:
        print(f"  {i:4d} taps {int(s[i,4]):3d} start {(s[i,6]-s[:,6].min())/100:6.1f} us  end {(s[i,7]-s[:,6].min())/100:6.1f} us  total {tot[i]:9.0f}  stage {s[i,1]-s[i,0]:9.0f}")
    for taps in sorted(set(s[:, 4].astype(int))):
        m = s[:, 4] == taps
        e = (s[m, 7] - s[:, 6].min()) / 100
        print(f"taps {taps:3d}: end time us p10 {np.percentile(e,10):6.1f} p50 {np.median(e):6.1f} p90 {np.percentile(e,90):6.1f} max {e.max():6.1f}; start max {((s[m,6]-s[:,6].min())/100).max():5.1f}")

if which == "conv":
    raw = buf.cpu().numpy().reshape(-1, 8)[(buf.cpu().numpy().reshape(-1, 8)[:, 6] > 0)]
    raw = raw[raw[:, 3] > 0]
    lo = (raw[:, 5] >> 32).astype(np.int64); hi = (raw[:, 5] & 0xffffffff).astype(np.int64)
    nrows = hi - lo
    tile0 = lo // B; tile1 = (hi - 1) // B
    tot = (raw[:, 3] - raw[:, 0]).astype(np.float64)
    for taps in sorted(set(raw[:, 4].astype(int))):
        m = raw[:, 4] == taps
        print(f"taps {taps}: rows/wg {nrows[m].min()}..{nrows[m].max()}")
        for t in sorted(set(tile0[m])):
            mm = m & (tile0 == t) & (tile1 == t)
            if mm.sum():
                print(f"    tile {t}: n={mm.sum():3d} cycles/row {np.mean(tot[mm]/nrows[mm]):8.0f}  total p50 {np.median(tot[mm]):9.0f}")
