#!/usr/bin/env python3
"""One TimesBlock forward timed with events at an arbitrary shape: python tools/block_time.py B L d_model [k]
(BASELINE configs[3] block: 256 720 128)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as ge

pkg = ge.load_package()
T = pkg.models.timesnet
B, L, C = (int(v) for v in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ks = [(3, 3), (5, 5), (7, 7)]
dev = torch.device("cuda:0")
params = pkg.synth.make_inception_params(C, 4 * C, ks, 4.0, seed=0)
blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
blk.period_selector = T.FFTPeriodSelector(K, L)
blk = blk.eval().to(dev)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
with torch.inference_mode():
    for _ in range(20):
        blk(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        blk(x)
    e1.record()
    torch.cuda.synchronize()
print(f"TimesBlock forward B={B} L={L} d_model={C} k={K}: {e0.elapsed_time(e1) / n:.3f} ms (backend {blk._last_backend}, engine {blk.check_range()})")
