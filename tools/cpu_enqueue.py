#!/usr/bin/env python3
"""Host time to ENQUEUE one TimesBlock forward vs the GPU time it takes (run on the GPU box).
FTN_BENCH_FORCE_DIST=1 adds the RCCL exchange at world size 1."""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge

pkg = ge.load_package()
T = pkg.models.timesnet
dev = torch.device("cuda:0")
B, L, C, K = 256, 336, 64, 5
ks = [(3, 3), (5, 5), (7, 7)]
blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
blk.period_selector = T.FFTPeriodSelector(K, L)
blk = blk.eval().to(dev)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
run = blk
if os.environ.get("FTN_BENCH_FORCE_DIST") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sh = pkg.dist.ShardedTimesBlock(blk)
    run = lambda t: sh(t, gather=False)
with torch.inference_mode():
    for _ in range(10):
        run(x)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        run(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if os.environ.get("STAGE_TIMING") == "1":
        lib = pkg.lib.load()
        lib.ftn_stage_timing(1)
        t0 = time.perf_counter()
        for _ in range(n):
            run(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        lib.ftn_stage_timing(0)
        print("with stage events:", end=" ")
print(f"host enqueue {1e6 * (t1 - t0) / n:.1f} us/step; total {1e6 * (t2 - t0) / n:.1f} us/step")
