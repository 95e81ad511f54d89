"""Drop-in check against the reference's OWN model shell (build container only: skipped
when /root/reference is absent, e.g. on the GPU box).  The mirrors replace
``FFTPeriodSelector`` / ``TimesBlock`` / ``LowRankTemporalContext`` inside the reference's
``timesnet_forecast.models.timesnet`` module; the reference ``TimesNet`` is then built,
loaded with ``load_state_dict(strict=True)`` from an unpatched twin (checkpoint ABI,
reference predict.py:687-722) and must produce the same ``[B,T,N] -> [B,H,N]`` outputs."""
import importlib
import sys
from pathlib import Path

import pytest
import torch

REF = Path("/root/reference/src")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")


def _ref_module():
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    return importlib.import_module("timesnet_forecast.models.timesnet")


def _build(mod, cfg):
    torch.manual_seed(0)
    model = mod.TimesNet(**cfg)
    return model


CFG = dict(input_len=48, pred_len=12, d_model=16, d_ff=32, n_layers=2, k_periods=3,
           kernel_set=[(3, 3), (5, 5)], dropout=0.0, activation="gelu", mode="direct",
           bottleneck_ratio=2.0, use_checkpoint=False, id_embed_dim=8, static_proj_dim=8,
           use_zero_mean_context=True, context_rank=4, context_scale=0.05)


def test_mirrors_drop_into_reference_timesnet(ftn, monkeypatch):
    ref = _ref_module()
    T = ftn.models.timesnet
    B, L, N = 3, 48, 5
    g = torch.Generator().manual_seed(1)
    t = torch.arange(L, dtype=torch.float32).view(1, L, 1)
    x = torch.randn(B, L, N, generator=g) + 2.0 * torch.sin(2 * torch.pi * t / 12.0) + torch.sin(2 * torch.pi * t / 8.0)
    static = torch.randn(N, 6, generator=g)
    ids = torch.arange(N)

    with torch.no_grad():
        orig = _build(ref, CFG).eval()
        orig(x, series_static=static, series_ids=ids)          # materialise lazy layers
        # non-zero heads, otherwise context/head paths are dead
        for p in orig.parameters():
            if float(p.detach().abs().sum()) == 0.0:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        want_rate, want_disp = orig(x, series_static=static, series_ids=ids)
        want_periods = orig.period_selector.last_selected_periods.tolist()

        monkeypatch.setattr(ref, "FFTPeriodSelector", T.FFTPeriodSelector)
        monkeypatch.setattr(ref, "TimesBlock", T.TimesBlock)
        monkeypatch.setattr(ref, "LowRankTemporalContext", T.LowRankTemporalContext)
        swapped = _build(ref, CFG).eval()
        swapped(x, series_static=static, series_ids=ids)
        assert type(swapped.blocks[0]) is T.TimesBlock and type(swapped.period_selector) is T.FFTPeriodSelector
        assert type(swapped.temporal_context) is T.LowRankTemporalContext
        assert set(swapped.state_dict().keys()) == set(orig.state_dict().keys())
        swapped.load_state_dict(orig.state_dict(), strict=True)
        rate, disp = swapped(x, series_static=static, series_ids=ids)

    assert swapped.period_selector.last_selected_periods.tolist() == want_periods
    assert all(b._period_calls >= 1 and b._last_backend == "torch" for b in swapped.blocks)
    torch.testing.assert_close(rate, want_rate, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(disp, want_disp, rtol=1e-5, atol=1e-6)
