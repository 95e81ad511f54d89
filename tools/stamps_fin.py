#!/usr/bin/env python3
"""Diagnostic: phase timeline of the finalize workgroup inside k_finalize_pw (s_memtime stamps, ftn_debug_stamps
which = 8).  Usage on the GPU box: python tools/stamps_fin.py [B L d_model]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as ge

pkg = ge.load_package()
lib = pkg.lib.load()
T = pkg.models.timesnet
dev = torch.device("cuda:0")
B, L, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 336, 64)
K = 5
ks = [(3, 3), (5, 5), (7, 7)]
params = pkg.synth.make_inception_params(C, 4 * C, ks, 4.0, seed=0)
blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
blk.period_selector = T.FFTPeriodSelector(K, L)
blk = blk.eval().to(dev)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
rows = []
with torch.inference_mode():
    for _ in range(3):
        blk(x)
    torch.cuda.synchronize()
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.ftn_debug_stamps(buf.data_ptr(), buf.numel(), 8)
    for _ in range(20):
        blk(x)
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy().copy())
    lib.ftn_debug_stamps(None, 0, 0)
s = np.array(rows, dtype=np.float64)
names = ["entry -> scores (psum wait, mean, penalty)", "top-k (one wave)", "periods / grouping / tiles (one wave)",
         "flagged grouping (off) + barrier", "descriptor write -> per-row weights + amps"]
d = np.diff(s[:, :6], axis=1)
for i, n in enumerate(names):
    print(f"{n:48s} median {np.median(d[:, i]):8.0f} ticks")
print(f"{'total':48s} median {np.median(s[:, 5] - s[:, 0]):8.0f} ticks")
