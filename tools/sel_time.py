#!/usr/bin/env python3
"""Selector front end (ftn_period_spectrum: DFT amplitudes, channel medians, batch sums) timed alone with events.
usage: python tools/sel_time.py [B L C ...]   (default: the bench shape and the BASELINE configs[3]/[4] shard shape)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as ge

ftn = ge.load_package()
args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(256, 336, 64), (64, 720, 128), (256, 720, 128)]
for B, L, C in shapes:
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=1, planted=(24, 7))).cuda()
    for _ in range(5):
        ftn.runtime.spectrum(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        ftn.runtime.spectrum(x)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} L={L} C={C}: {e0.elapsed_time(e1) / n * 1000:.1f} us per ftn_period_spectrum call", flush=True)
