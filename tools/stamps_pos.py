#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of the position-major stage C (k_mlp_pos) from s_memtime stamps
(ftn_debug_stamps).  Usage on the GPU box: python tools/stamps_pos.py   (STAMP_B = batch rows)
slots: 0 start, 1 first chunk top, [chunk 1 of pass 0:] 2 top, 3 weights landed + barrier, 4 compute done, 5 end barrier,
6 pass 0 stored, 7 kernel end"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge

pkg = ge.load_package()
lib = pkg.lib.load()
T = pkg.models.timesnet
dev = torch.device("cuda:0")
B, L, C, K = int(os.environ.get("STAMP_B", "256")), 336, 64, 5
ks = [(3, 3), (5, 5), (7, 7)]
params = pkg.synth.make_inception_params(C, 4 * C, ks, 4.0, seed=0)
blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
blk.period_selector = T.FFTPeriodSelector(K, L)
blk = blk.eval().to(dev)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
with torch.inference_mode():
    for _ in range(30):
        blk(x)
    torch.cuda.synchronize()
    buf = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    lib.ftn_debug_stamps(buf.data_ptr(), buf.numel(), 2)
    blk(x)
    torch.cuda.synchronize()
    lib.ftn_debug_stamps(None, 0, 0)
s = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 7] > 0]
print("workgroups that ran to the end:", len(s))
t0 = s[:, 0].min()
# s_memtime ticks at 100 MHz on gfx950 (constant-rate counter)
names = [("prologue (decode, x, biases)", 0, 1), ("chunk 1: refill + barrier", 2, 3), ("chunk 1: compute", 3, 4), ("chunk 1: closing barrier", 4, 5),
         ("pass 0 (all chunks + a' stores)", 1, 6), ("rest (further passes, R store)", 6, 7), ("total", 0, 7)]
for name, a, b in names:
    dt = (s[:, b] - s[:, a]) * 10.0          # ns
    print(f"{name:34s} mean {dt.mean()/1e3:8.2f} us  p50 {np.median(dt)/1e3:8.2f}  p90 {np.percentile(dt, 90)/1e3:8.2f}  max {dt.max()/1e3:8.2f}")
st = (s[:, 0] - t0) / 100.0
en = (s[:, 7] - t0) / 100.0
print("kernel span %.1f us" % en.max())
edges = np.linspace(0, en.max(), 17)
print("resident workgroups at t:", [int(((st <= e) & (en > e)).sum()) for e in edges[:-1]])
