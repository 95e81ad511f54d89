"""torch.profiler table of one whole-model forward (eager) - which torch ops the shell still issues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
B, L, H, N, D = 256, 336, 96, 512, 64
ks = [(3, 3), (5, 5), (7, 7)]
cfg = dict(input_len=L, pred_len=H, d_model=D, d_ff=4 * D, n_layers=3, k_periods=5, kernel_set=ks, dropout=0.0,
           activation="gelu", mode="direct", bottleneck_ratio=4.0, id_embed_dim=32, use_zero_mean_context=True, context_rank=16)
torch.manual_seed(0)
m = pkg.models.TimesNet(**cfg).eval()
x = torch.randn(B, L, N)
with torch.no_grad():
    m(x[:2])
m = m.to(dev); x = x.to(dev)
from torch.profiler import profile, ProfilerActivity
with torch.inference_mode():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        m(x); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
