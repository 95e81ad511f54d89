#!/usr/bin/env python3
"""Selector-only loop for profiling (run on the GPU box): python tools/sel_bench.py [B L C K iters]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge

a = [int(v) for v in sys.argv[1:]]
B, L, C, K, iters = (a + [256, 336, 64, 5, 30][len(a):])[:5]
pkg = ge.load_package()
T = pkg.models.timesnet
dev = torch.device("cuda:0")
sel = T.FFTPeriodSelector(K, L)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
with torch.inference_mode():
    for _ in range(3):
        sel.select_device(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        sel.select_device(x)
    e1.record()
    torch.cuda.synchronize()
print(f"selector B={B} L={L} C={C} k={K}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per call; periods",
      sel.last_selected_periods.tolist())
