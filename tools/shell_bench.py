"""Time the model-shell kernels (head, value embedding) at the headline shape on cuda:0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
rt = pkg.runtime
dev = torch.device("cuda:0")
B, L, N, D, S = 256, 336, 512, 64, 96


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(B, L, N, device=dev)
hidden = torch.randn(B, S, D, device=dev)
wmu, wsg = torch.randn(N, D, device=dev) * 0.1, torch.randn(N, D, device=dev) * 0.1
bmu, bsg = torch.randn(N, device=dev), torch.randn(N, device=dev)
late = torch.randn(1, S, N, device=dev)
us = timeit(lambda: rt.head_forward(hidden, wmu, bmu, wsg, bsg, x[:, -S:, :], S, late, None, 1e-3))
byts = 2 * B * S * N * 4 + B * S * N * 4 + B * S * D * 4
print("k_head   %.1f us  %.0f GB/s algorithmic (%.1f MB)" % (us, byts / us / 1e3, byts / 1e6))
w = torch.randn(D, N, device=dev) * 0.05
add = torch.randn(1, L, D, device=dev)
us = timeit(lambda: rt.embed_forward(x, w, add, None))
byts = B * L * N * 4 + B * L * D * 4
print("k_embed  %.1f us  %.0f GB/s algorithmic (%.1f MB), %.1f TFLOP/s fp32" % (us, byts / us / 1e3, byts / 1e6, 2.0 * B * L * N * D / us / 1e6))
