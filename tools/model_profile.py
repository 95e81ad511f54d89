"""Run only bench.model_bench (whole TimesNet.forward) - used under rocprofv3 to split the shell's
kernels from the block kernels, and to time other shapes.
usage: python3 tools/model_profile.py [iters] [B L N d_model]   (default: 10 256 336 512 64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, L, N, D = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 336, 512, 64)
ks = [(3, 3), (5, 5), (7, 7)]
r = bench.model_bench(pkg, dev, B, L, N, D, ks, 4.0, 5, iters=iters)
print(r)
