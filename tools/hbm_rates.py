"""What a plain streaming kernel reaches on this box: fill (write-only) and copy (read + write) of 176 MB."""
import torch
n = 256 * 336 * 512
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
us = t(lambda: a.fill_(1.0)); print("fill  %.1f us  %.0f GB/s" % (us, n * 4 / us / 1e3))
us = t(lambda: b.copy_(a)); print("copy  %.1f us  %.0f GB/s (read+write)" % (us, 2 * n * 4 / us / 1e3))
us = t(lambda: torch.add(a, b, out=b)); print("add   %.1f us  %.0f GB/s (2 reads + write)" % (us, 3 * n * 4 / us / 1e3))
us = t(lambda: a.sum()); print("sum   %.1f us  %.0f GB/s (read)" % (us, n * 4 / us / 1e3))
