import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); rt = pkg.runtime
dev = torch.device("cuda:0")
B, L, C = 256, 336, 64
ks = [(3,3),(5,5),(7,7)]
blk = pkg.models.TimesBlock(C, ks, 0.0, "gelu", d_ff=4*C, bottleneck_ratio=4.0).eval().to(dev)
blk.period_selector = pkg.models.FFTPeriodSelector(5, L)
x = torch.from_numpy(pkg.synth.make_input(B, L, C, seed=0)).to(dev)
wblob, plan = blk._packed(dev)
sm = blk.period_selector
for _ in range(5): rt.stage_a_only(x, plan, wblob, sm.k, sm.pmax, sm.min_period_threshold)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): rt.stage_a_only(x, plan, wblob, sm.k, sm.pmax, sm.min_period_threshold)
e1.record(); torch.cuda.synchronize()
print("stage A only: %.1f us" % (e0.elapsed_time(e1) / 200 * 1e3))
